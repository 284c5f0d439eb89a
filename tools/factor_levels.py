#!/usr/bin/env python3
"""Per tree level of the numeric factorisation (cfg 2): wall span of the level, and per kernel launches / summed kernel time /
time on the critical path is NOT derived -- the table shows which kernels fill a level's span and how much of it nothing covers.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fl_trace -- python3 tools/factor_levels.py run [grid]
    python3 tools/factor_levels.py join gpurun_out/fl_trace

The library is run with GMRFX_LEVEL_MARK=1: an empty marker kernel (k_level_mark, phase 3 = factorisation, grid = level + 2) is
launched on the main stream at the top of every level."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(grid):
    os.environ["GMRFX_LEVEL_MARK"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
    import numpy as np, torch
    import gmrfx
    from gmrfx import spde
    mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
    Q = spde.matern_precision(mesh, 0, 0.2)
    dev = torch.device("cuda", 0)
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
    torch.cuda.synchronize()
    for _ in range(3):
        be.refactorize_dev(d_nz.data_ptr())
    torch.cuda.synchronize()
    print("ms_factor", be.stats()["ms_factor"])


def nm(r):
    return r["Kernel_Name"].split("(")[0].replace("gmrfx::", "").replace("void ", "")


def join(d):
    f = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the LAST factorisation: from the last marker of level 0 (grid = 2 workgroups of 192 threads) on
    marks = [i for i, r in enumerate(rows) if nm(r) == "k_level_mark" and int(r["Workgroup_Size_X"]) == 192]
    first = [i for i in marks if int(rows[i]["Grid_Size_X"]) // 192 - 2 == 0][-1]
    # subtree kernels precede the first level marker: walk back to the previous k_level_mark or 40 kernels
    lo = first
    while lo > 0 and first - lo < 40 and nm(rows[lo - 1]) != "k_level_mark":
        lo -= 1
    cur, groups, order = -1, {}, []
    for r in rows[lo:]:
        if nm(r) == "k_level_mark":
            if int(r["Workgroup_Size_X"]) != 192:
                continue
            cur = int(r["Grid_Size_X"]) // 192 - 2
            t_mark = int(r["Start_Timestamp"])
            groups.setdefault(cur, {"t0": t_mark, "k": []})
            order.append(cur)
            continue
        groups.setdefault(cur, {"t0": int(r["Start_Timestamp"]), "k": []})["k"].append(r)
    lv = sorted(groups)
    print("level   span us   covered us (union)   per kernel: launches x avg us = sum us")
    tot_span = 0.0
    for a, l in enumerate(lv):
        g = groups[l]
        if not g["k"]:
            continue
        t0 = g["t0"]
        t1 = groups[lv[a + 1]]["t0"] if a + 1 < len(lv) else max(int(r["End_Timestamp"]) for r in g["k"])
        span = (t1 - t0) / 1e3
        tot_span += span
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in g["k"])
        cov, ce = 0, iv[0][0]
        for s, e in iv:
            s = max(s, ce)
            if e > s:
                cov += e - s
                ce = e
        names = {}
        for r in g["k"]:
            n_ = nm(r)
            names.setdefault(n_, [0, 0.0, 0])
            names[n_][0] += 1
            names[n_][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            names[n_][2] = max(names[n_][2], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
        parts = ", ".join(f"{k} {v[0]}x{v[1] / v[0]:.1f}={v[1]:.0f} (<= {v[2]} wg)" for k, v in sorted(names.items(), key=lambda kv: -kv[1][1]))
        print(f"{l:5d} {span:9.1f} {cov / 1e3:9.1f}   {parts}")
    print(f"sum of spans {tot_span / 1e3:.3f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
    else:
        join(sys.argv[2])
