#!/usr/bin/env python3
"""Sweep the symbolic-analysis knobs (nested-dissection leaf size, relaxed amalgamation) on the
headline workload and print factor / solve times per setting. Needs a GPU.
usage: python tools/sweep_opts.py [grid] [nrhs]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, smoothness=0, range_=0.2)
n = Q.shape[0]
dev = torch.device("cuda", 0)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
d_B = torch.randn((nrhs, n), dtype=torch.float64, generator=torch.Generator().manual_seed(1)).to(dev)
d_X = torch.empty_like(d_B)
cfgs = [dict()]
for rc in (16, 48, 64):
    cfgs.append(dict(relax_cols=rc))
for rz in (0.08, 0.25, 0.35):
    cfgs.append(dict(relax_zeros=rz))
for nl in (32, 96, 128):
    cfgs.append(dict(nd_leaf=nl))
cfgs += [dict(relax_cols=48, relax_zeros=0.25), dict(relax_cols=64, relax_zeros=0.35), dict(nd_leaf=96, relax_cols=48, relax_zeros=0.25)]
for cfg in cfgs:
    t0 = time.perf_counter()
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False, **cfg)
    tsym = time.perf_counter() - t0
    fs, ss = [], []
    for it in range(4):
        be.refactorize_dev(d_nz.data_ptr())
        be.solve_dev(d_B.data_ptr(), n, nrhs, d_X.data_ptr(), n)
        s = be.stats()
        if it:
            fs.append(s["ms_factor"]); ss.append(s["ms_solve"])
    st = be.stats()
    print(json.dumps({"cfg": cfg, "ms_factor": round(min(fs), 2), "ms_solve": round(min(ss), 2), "sum": round(min(fs) + min(ss), 2),
                      "nsuper": st.get("nsuper"), "nnz_L": st.get("nnz_factor", st.get("nnz_L")), "flops": st.get("flops"),
                      "sym_s": round(tsym, 2)}), flush=True)
    be.close()
