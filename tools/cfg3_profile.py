#!/usr/bin/env python3
"""BASELINE cfg 3 evidence: the Takahashi selected inversion and the 256-sample backward solve on the cfg-2 factor, per kernel
(time from a rocprofv3 kernel trace, HBM bytes from two PMC passes) and, for the selected inversion, per tree level.

    cd /tmp && export TMPDIR=/tmp && cd -
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3_trace -- python3 tools/cfg3_profile.py run
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/c3_fetch -- python3 tools/cfg3_profile.py run
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/c3_write -- python3 tools/cfg3_profile.py run
    python3 tools/cfg3_profile.py join gpurun_out/c3_trace gpurun_out/c3_fetch gpurun_out/c3_write profiles/r05_cfg3

`run` factorises, then does the selected inversion and the 256-sample backward solve twice; `join` cuts the LAST of each out of
the traces (phases: everything between the last factorisation kernel and the next k_permute = selected inversion; from there to
the end = the samples; levels: the empty marker kernel the library launches per level under GMRFX_LEVEL_MARK=1, grid = level + 2).
FETCH_SIZE is doubled (gfx950 reports half of the wide coalesced reads; MI355X_MICROARCH.md). Writes <prefix>_kernel_stats.csv,
<prefix>_pmc_traffic.json and <prefix>_selinv_levels.txt."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
META = os.path.join(ROOT, "gpurun_out", "cfg3_meta.json")


def run():
    os.environ["GMRFX_LEVEL_MARK"] = "1"
    import numpy as np
    import torch
    import gmrfx
    from gmrfx import spde
    grid = int(os.environ.get("CFG3_GRID", "1000"))
    mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
    Q = spde.matern_precision(mesh, smoothness=0, range_=0.2)
    n = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
    dev = torch.device("cuda", 0)
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
    d_Z = torch.randn((256, n), generator=torch.Generator(device="cpu").manual_seed(2), dtype=torch.float64).to(dev)
    d_S = torch.empty_like(d_Z)
    ms = []
    for _ in range(2):
        be.refactorize_dev(d_nz.data_ptr())
        be.selinv_compute_dev()
        t_sel = be.stats()["ms_selinv"]
        be.backward_solve_dev(d_Z.data_ptr(), n, 256, d_S.data_ptr(), n)
        ms.append((t_sel, be.stats()["ms_backward_solve"]))
    torch.cuda.synchronize()
    sy = be.symbolic()
    c = np.diff(sy.super_first).astype(float)
    m = np.diff(sy.row_ptr).astype(float) - c
    st = be.stats()
    levels = []
    for lv in range(int(sy.level.max()) + 1):
        sel = sy.level == lv
        cc, mm = c[sel], m[sel]
        levels.append({"level": lv, "fronts": int(sel.sum()), "c_max": int(cc.max()) if sel.any() else 0,
                       "flops": float((2 * cc * mm * mm + 4 * cc * cc * mm + 2 * cc ** 3 / 3).sum()),
                       "bytes_min": float((16 * (cc + mm) * cc).sum())})          # panel of L read + panel of Z written
    os.makedirs(os.path.dirname(META), exist_ok=True)
    json.dump({"grid": grid, "n": n, "ms_selinv": ms[-1][0], "ms_rand256": ms[-1][1], "nnz_l_stored": st["nnz_l_stored"],
               "sum_rows": st["sum_rows"], "levels": levels}, open(META, "w"))
    print("ms_selinv, ms_rand256:", ms)


def short(name):
    return name.split("(")[0].replace("gmrfx::", "").replace("void ", "").strip()


def is_factor(n):
    return n.startswith(("k_factor", "k_assemble", "k_potrf", "k_gemm_nt", "k_syrk")) or n.startswith("k_trsm<0")


def cut(names):
    """(first index of the last selected inversion, first index of the last sample batch, end)"""
    last_f = max(i for i, n in enumerate(names) if is_factor(n))
    sel0 = last_f + 1
    perm = [i for i, n in enumerate(names) if n.startswith("k_permute") and i > sel0]
    return sel0, perm[0], len(names)


def load_trace(d):
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)) for r in rows]


def load_pmc(d):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(short(r["Kernel_Name"]), float(r["Counter_Value"]) * 1024.0) for r in rows]


def join(trace_d, fetch_d, write_d, prefix):
    meta = json.load(open(META))
    T = load_trace(trace_d)
    names = [t[0] for t in T]
    s0, s1, s2 = cut(names)
    F, W = load_pmc(fetch_d), load_pmc(write_d)
    f0, f1, f2 = cut([x[0] for x in F])
    w0, w1, w2 = cut([x[0] for x in W])
    out = {"workload": {"grid": meta["grid"], "n": meta["n"], "samples": 256},
           "note": "rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE in three separate runs of tools/cfg3_profile.py run; "
                   "FETCH_SIZE x2 (gfx950); the last selected inversion and the last 256-sample backward solve of the run"}
    stats_rows = []
    for ph, (a, b), (fa, fb), (wa, wb) in (("selinv", (s0, s1), (f0, f1), (w0, w1)), ("rand256", (s1, s2), (f1, f2), (w1, w2))):
        per = collections.OrderedDict()
        for n, us, _, _ in T[a:b]:
            if n.startswith("k_level_mark"):
                continue
            e = per.setdefault(n, {"calls": 0, "us": 0.0, "fetch_x2": 0.0, "write": 0.0})
            e["calls"] += 1; e["us"] += us
        for n, v in F[fa:fb]:
            if n in per:
                per[n]["fetch_x2"] += 2.0 * v
        for n, v in W[wa:wb]:
            if n in per:
                per[n]["write"] += v
        tot_us = sum(e["us"] for e in per.values())
        tot_b = sum(e["fetch_x2"] + e["write"] for e in per.values())
        out[ph] = {"kernel_us": tot_us, "fetch_bytes": sum(e["fetch_x2"] for e in per.values()), "write_bytes": sum(e["write"] for e in per.values()),
                   "total_bytes": tot_b, "per_kernel": {k: {"calls": e["calls"], "us": round(e["us"], 1), "GB": round((e["fetch_x2"] + e["write"]) / 1e9, 3)}
                                                        for k, e in sorted(per.items(), key=lambda kv: -kv[1]["us"])}}
        for k, e in sorted(per.items(), key=lambda kv: -kv[1]["us"]):
            stats_rows.append([ph, k, e["calls"], round(e["us"], 1), round(e["us"] / e["calls"], 1), round(100 * e["us"] / tot_us, 1),
                               round((e["fetch_x2"] + e["write"]) / 1e9, 3)])
    out["selinv"]["ms_event"] = meta["ms_selinv"]; out["rand256"]["ms_event"] = meta["ms_rand256"]
    from gmrfx._lib import source_tree_hash
    out["csrc_hash"] = source_tree_hash()       # the source tree these counters belong to (bench.py checks it)
    json.dump(out, open(prefix + "_pmc_traffic.json", "w"), indent=1)
    with open(prefix + "_kernel_stats.csv", "w") as fh:
        w = csv.writer(fh)
        w.writerow(["phase", "kernel", "calls", "total_us", "avg_us", "percent_of_phase", "hbm_GB"])
        w.writerows(stats_rows)
    # ---- per level of the selected inversion (top-down): marker kernels with 256 threads (phase 4), grid = level + 2
    lv_t = collections.OrderedDict()
    cur = None
    for n, us, gx, wx in T[s0:s1]:
        if n.startswith("k_level_mark"):
            cur = gx // max(wx, 1) - 2 if wx == 256 else cur
            if wx == 256:
                lv_t.setdefault(cur, collections.OrderedDict())
            continue
        if cur is not None:
            lv_t[cur][n] = lv_t[cur].get(n, 0.0) + us
    lv_b = collections.defaultdict(float)
    for L_, key in ((F[f0:f1], 2.0), (W[w0:w1], 1.0)):
        cur = None
        marks = iter([lv for lv in lv_t])
        for n, v in L_:
            if n.startswith("k_level_mark"):
                cur = next(marks, cur)
                continue
            if cur is not None:
                lv_b[cur] += key * v
    lines = ["== selected inversion, top-down: level, fronts, widest front, GFLOP (sum_s 2 c m^2 + 4 c^2 m + 2 c^3 / 3), minimal MB (L read + Z written), "
             "kernel us, TFLOP/s, PMC HBM MB, PMC / minimal, kernels",
             " level fronts c_max    GFLOP   min MB       us  TFLOP/s   PMC MB  ratio   kernels"]
    info = {l["level"]: l for l in meta["levels"]}
    tf = tu = 0.0
    for lv, ks in lv_t.items():
        us = sum(ks.values()); li = info.get(lv, {"fronts": 0, "c_max": 0, "flops": 0.0, "bytes_min": 0.0})
        tf += li["flops"]; tu += us
        kk = " ".join(f"{k}:{v:.0f}" for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:5])
        lines.append(f"{lv:6d} {li['fronts']:6d} {li['c_max']:5d} {li['flops'] / 1e9:8.2f} {li['bytes_min'] / 1e6:8.1f} {us:8.1f} {li['flops'] / max(us, 1e-9) / 1e6:8.1f} "
                     f"{lv_b[lv] / 1e6:8.1f} {lv_b[lv] / max(li['bytes_min'], 1.0):6.2f}   {kk}")
    lines.append(f" total: {tf / 1e9:.1f} GFLOP in {tu:.0f} us of kernel time = {tf / max(tu, 1e-9) / 1e6:.1f} TFLOP/s; HIP-event time of the call {meta['ms_selinv']:.2f} ms")
    open(prefix + "_selinv_levels.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-12:]))
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_kernel"} for k, v in out.items() if k in ("selinv", "rand256")}))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        join(*sys.argv[2:6])
