#!/usr/bin/env python3
"""Per-level table of the triangular sweeps at cfg 2 (the counterpart of tools/syrk_levels.py for the solve): for the
sweep tasks and for every tree level of the forward and the backward sweep -- fronts, algorithmic bytes, kernel time
(sum over the level's launches) and GB/s; with two counter passes also the HBM bytes the level really moved.

The library is run with GMRFX_LEVEL_MARK=1: an empty marker kernel (k_level_mark, launch geometry = phase and level)
precedes every level, so the trace is cut into levels without guessing launch orders.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/swl -- python3 tools/sweep_levels.py run
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/swl_f --output-format csv -- python3 tools/sweep_levels.py run
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/swl_w --output-format csv -- python3 tools/sweep_levels.py run
    python3 tools/sweep_levels.py join gpurun_out/swl [gpurun_out/swl_f gpurun_out/swl_w]

Algorithmic bytes of a level (per front of c columns, r rows, nrhs right-hand sides): the panel 8 r c, its
row list 4 r, its own rows of X read and written once 16 c nrhs. The W / x hand-off between a front and its parent
(8 nrhs (r - c) written and read once in the forward sweep; the trailing x gathered in the backward one) is what the
multifrontal schedule adds on top: listed separately as `handoff`.
"""
import csv, glob, json, os, sys
sys.path.insert(0, "gaussianmarkovrandomfields.jl_amd")
OUT = "gpurun_out/sweep_levels.json"
NRHS = int(os.environ.get("SWL_NRHS", "64"))      # SWL_NRHS=1: the single right-hand side solve (with SWL_NOMARK=1: no marker kernels)


def run():
    if not os.environ.get("SWL_NOMARK"): os.environ["GMRFX_LEVEL_MARK"] = "1"
    import numpy as np, torch, gmrfx
    from gmrfx import spde
    grid = int(os.environ.get("SWL_GRID", "1000"))
    mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
    Q = spde.matern_precision(mesh, 0, 0.2)
    n = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
    sy = be.symbolic()
    c = np.diff(sy.super_first).astype(float); r = np.diff(sy.row_ptr).astype(float); m = r - c
    lev = np.asarray(sy.level)
    cap, tf, tl, _ = be.sweep_tasks()
    in_task = np.zeros(len(c), bool)
    for a, b in zip(tf, tl):
        in_task[a:b + 1] = True
    roots = np.zeros(len(c), bool); roots[tl] = True

    def entry(sel, name):
        return dict(level=name, fronts=int(sel.sum()), c_max=int(c[sel].max()) if sel.any() else 0,
                    cols=float(c[sel].sum()),
                    bytes=float((8 * r[sel] * c[sel] + 4 * r[sel] + 16 * c[sel] * NRHS).sum()),
                    handoff_out=float((8 * NRHS * m[sel]).sum()))
    levels = {}
    e = entry(in_task, -1)
    e["handoff_out"] = float((8 * NRHS * m[roots]).sum())      # only the task roots hand anything to HBM
    e["tasks"] = int(len(tf))
    levels[-1] = e
    kids_of = {}
    par = np.asarray(sy.super_parent)
    for lv in range(int(lev.max()) + 1):
        sel = (~in_task) & (lev == lv)
        e = entry(sel, lv)
        kid = np.isin(par, np.nonzero(sel)[0]) & (par >= 0)
        # children's update vectors read by this level; a task's inner fronts hand off inside LDS
        kid_out = kid & (~in_task | roots)
        e["handoff_in"] = float((8 * NRHS * m[kid_out]).sum())
        levels[lv] = e
    json.dump({"n": int(n), "nrhs": NRHS, "levels": [levels[k] for k in sorted(levels)]}, open(OUT, "w"))
    dev = torch.device("cuda:0")
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
    d_B = torch.randn((NRHS, n), generator=torch.Generator(device="cpu").manual_seed(1), dtype=torch.float64).to(dev)
    d_X = torch.empty_like(d_B)
    torch.cuda.synchronize()
    be.refactorize_dev(d_nz.data_ptr())
    for _ in range(3):              # the first solve after a refactorisation also computes the dense inverses
        be.solve_dev(d_B.data_ptr(), n, NRHS, d_X.data_ptr(), n)
    torch.cuda.synchronize()
    st = be.stats()
    print(json.dumps({"ms_solve_fwd": st["ms_solve_fwd"], "ms_solve_bwd": st["ms_solve_bwd"]}))


def nm(r):
    return r["Kernel_Name"].split("(")[0].replace("gmrfx::", "").replace("void ", "")


def cut(rows, key):
    """kernels of the LAST solve (between its two k_permute launches), grouped by (phase, level) marker"""
    idx = [i for i, r in enumerate(rows) if nm(r).startswith("k_permute")]
    a, b = idx[-2], idx[-1]
    groups, cur = {}, None
    for r in rows[a + 1:b]:
        if nm(r) == "k_level_mark":
            wg = int(r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r["Workgroup_Size"])     # (counter passes: one column)
            gr = int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"])
            phase, level = wg // 64, gr // wg - 2
            cur = (phase, level)
            groups.setdefault(cur, [])
            continue
        if cur is not None:
            groups[cur].append(r)
    return groups


def join(d, dfetch=None, dwrite=None):
    meta = json.load(open(OUT))
    levels = {l["level"]: l for l in meta["levels"]}
    f = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    g = cut(rows, None)
    traffic = None
    if dfetch and dwrite:
        def load(dd):
            ff = sorted(glob.glob(dd + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
            rr = list(csv.DictReader(open(ff)))
            rr.sort(key=lambda r: int(r["Dispatch_Id"]))
            gg = cut(rr, None)
            return {k: sum(float(x["Counter_Value"]) for x in v) * 1024.0 for k, v in gg.items()}
        F, W = load(dfetch), load(dwrite)
        traffic = {k: 2.0 * F.get(k, 0.0) + W.get(k, 0.0) for k in set(F) | set(W)}      # FETCH_SIZE x2 on gfx950
    tot = {1: [0.0, 0.0, 0.0, 0.0], 2: [0.0, 0.0, 0.0, 0.0]}
    for phase, title in ((1, "FORWARD sweep"), (2, "BACKWARD sweep")):
        print(f"== {title}: level (-1 = the LDS sweep tasks), fronts, widest front, algorithmic MB (panels + rows + own X), hand-off MB "
              f"(update vectors through HBM), kernel us (sum of the level's launches), GB/s on the algorithmic bytes"
              + (", PMC HBM MB, PMC / (algorithmic + hand-off)" if traffic else ""))
        print(" level fronts c_max    alg MB  handoff MB  launches      us    GB/s" + ("   PMC MB  ratio" if traffic else "") + "   kernels")
        keys = sorted([k for k in g if k[0] == phase], key=lambda k: k[1], reverse=(phase == 2))
        for k in keys:
            lv = levels.get(k[1])
            if lv is None:
                continue
            ks = g[k]
            us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks) / 1e3
            ho = lv["handoff_out"] + lv.get("handoff_in", 0.0) if phase == 1 else lv["handoff_out"]
            names = {}
            for r in ks:
                names[nm(r)] = names.get(nm(r), 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            ktxt = " ".join(f"{a}:{b:.0f}" for a, b in sorted(names.items(), key=lambda kv: -kv[1]))
            line = (f"{k[1]:6d} {lv['fronts']:6d} {lv['c_max']:5d} {lv['bytes']/1e6:9.1f} {ho/1e6:11.1f} {len(ks):9d} {us:7.1f} "
                    f"{(lv['bytes']/us/1e3 if us > 0 else 0):7.0f}")
            if traffic:
                t = traffic.get(k, 0.0)
                line += f" {t/1e6:8.1f} {t/max(lv['bytes'] + ho, 1.0):6.2f}"
                tot[phase][3] += t
            print(line + "   " + ktxt)
            tot[phase][0] += lv["bytes"]; tot[phase][1] += ho; tot[phase][2] += us
        a, h, u, t = tot[phase]
        print(f" total: algorithmic {a/1e9:.3f} GB + hand-off {h/1e9:.3f} GB in {u:.0f} us of kernel time = {a/u/1e3:.0f} GB/s algorithmic"
              + (f"; PMC {t/1e9:.3f} GB = {t/a:.2f} x algorithmic" if traffic else ""))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        join(*sys.argv[2:5])
