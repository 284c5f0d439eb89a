#!/usr/bin/env python3
"""Per-level table of the contribution-block SYRK (k_syrk_cb) at cfg 2: fronts, tiles, flops and -- when a
rocprofv3 kernel trace of THIS script is given -- the duration and TFLOP/s of each launch.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/syl -- python3 tools/syrk_levels.py run
    python3 tools/syrk_levels.py join gpurun_out/syl

and, from two separate counter passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE ... -- python3 tools/syrk_levels.py run),
the L2-miss traffic of each launch next to its algorithmic bytes (FETCH_SIZE doubled on gfx950, see tools/pmc_traffic.py):

    python3 tools/syrk_levels.py traffic gpurun_out/syl_fetch gpurun_out/syl_write
"""
import csv, glob, json, os, sys
sys.path.insert(0, "gaussianmarkovrandomfields.jl_amd")
OUT = "gpurun_out/syrk_levels.json"


def run():
    import numpy as np, torch, gmrfx
    from gmrfx import spde
    mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
    Q = spde.matern_precision(mesh, 0, 0.2)
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
    sy = be.symbolic()
    c = np.diff(sy.super_first); r = np.diff(sy.row_ptr); m = r - c
    small = int(os.environ.get("GMRFX_SMALL_ROWS", "64"))
    big = (r > small) & (m > 0)
    levels = []
    for lv in range(int(sy.level.max()) + 1):
        sel = big & (sy.level == lv)
        if not sel.any():
            continue
        cc, mm = c[sel].astype(float), m[sel].astype(float)
        t = np.ceil(mm / 64)
        kids = np.isin(sy.super_parent, np.nonzero(sel)[0])
        byts = float((mm * cc * 8).sum() + (mm * (mm + 1) / 2 * 8).sum() + (m[kids] * (m[kids] + 1.0) / 2 * 8).sum())
        levels.append(dict(level=lv, fronts=int(sel.sum()), c_max=int(cc.max()), m_max=int(mm.max()), bytes=byts,
                           tiles=int((t * (t + 1) / 2).sum()), flops=float((cc * mm * (mm + 1)).sum()),
                           tile_flops=float((cc * 2 * 64 * 64 * t * (t + 1) / 2).sum())))
    json.dump(levels, open(OUT, "w"))
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to("cuda:0")
    for _ in range(3):
        be.refactorize_dev(d_nz.data_ptr())
    torch.cuda.synchronize()


def join(d):
    levels = json.load(open(OUT))
    f = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if "k_syrk_cb" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(levels)
    rows = rows[-n:]                      # the last refactorisation; launches run bottom level first
    tot_t = tot_f = 0.0
    print("level fronts  c_max  m_max   tiles   GFLOP  tile-GFLOP      us   TFLOP/s  (useful / issued)")
    for lv, r in zip(levels, rows):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot_t += us; tot_f += lv["flops"]
        print(f"{lv['level']:5d} {lv['fronts']:6d} {lv['c_max']:6d} {lv['m_max']:6d} {lv['tiles']:7d} {lv['flops']/1e9:7.2f} "
              f"{lv['tile_flops']/1e9:11.2f} {us:7.1f} {lv['flops']/us/1e6:9.1f} / {lv['tile_flops']/us/1e6:6.1f}")
    print(f"total {tot_f/1e9:.1f} GFLOP in {tot_t:.0f} us = {tot_f/tot_t/1e6:.1f} TFLOP/s")


def traffic(dfetch, dwrite):
    levels = json.load(open(OUT))
    def load(d):
        f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
        rows = [r for r in csv.DictReader(open(f)) if "k_syrk_cb" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        return [float(r["Counter_Value"]) * 1024.0 for r in rows[-len(levels):]]
    F, W = load(dfetch), load(dwrite)
    print("level fronts  algorithmic MB   fetched MB (x2)   written MB   ratio")
    tf = tw = ta = 0.0
    for lv, f, w in zip(levels, F, W):
        f *= 2.0
        tf += f; tw += w; ta += lv["bytes"]
        print(f"{lv['level']:5d} {lv['fronts']:6d} {lv['bytes']/1e6:15.1f} {f/1e6:17.1f} {w/1e6:12.1f} {(f + w)/lv['bytes']:7.2f}")
    print(f"total: algorithmic {ta/1e9:.2f} GB, fetched {tf/1e9:.2f} GB, written {tw/1e9:.2f} GB")


def pmc(dirs):
    """per-launch values of whatever counters the given rocprofv3 --pmc output directories hold"""
    levels = json.load(open(OUT))
    cols = {}
    for d in dirs:
        f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
        rows = [r for r in csv.DictReader(open(f)) if "k_syrk_cb" in r["Kernel_Name"]]
        by = {}
        for r in rows:
            by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for name, v in by.items():
            v.sort()
            cols[name] = [x for _, x in v[-len(levels):]]
    names = sorted(cols)
    print("level " + " ".join(f"{n[-24:]:>24s}" for n in names))
    for k, lv in enumerate(levels):
        print(f"{lv['level']:5d} " + " ".join(f"{cols[n][k]:24.4g}" for n in names))


if __name__ == "__main__":
    if sys.argv[1] == "pmc": pmc(sys.argv[2:])
    elif sys.argv[1] == "run": run()
    elif sys.argv[1] == "join": join(sys.argv[2])
    else: traffic(sys.argv[2], sys.argv[3])
