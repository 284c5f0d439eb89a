#!/usr/bin/env python3
"""Capacity check: a larger 2-D problem than the headline one (default 2000 x 2000 = 4M nodes) and a 3-D one,
one refactorise + 64-RHS solve each, residual and timings. usage: big_run.py [grid2d] [grid3d]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde

def run(name, mesh, Q, nrhs=64):
    n = Q.shape[0]
    t0 = time.time()
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
    tsym = time.time() - t0
    dev = torch.device("cuda", 0)
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
    Bh = torch.randn((nrhs, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    d_B = Bh.to(dev); d_X = torch.empty_like(d_B)
    for _ in range(2):
        be.refactorize_dev(d_nz.data_ptr())
        be.solve_dev(d_B.data_ptr(), n, nrhs, d_X.data_ptr(), n)
    st = be.stats()
    X = d_X.cpu().numpy().T
    resid = float(np.linalg.norm(Q @ X - Bh.numpy().T) / np.linalg.norm(Bh.numpy()))
    print(f"{name}: n={n} nnz(L)={st['nnz_l']:.3e} flops={st['factor_flops']:.3e} symbolic {tsym:.1f}s factor {st['ms_factor']:.1f} ms "
          f"({st['factor_flops']/st['ms_factor']/1e9:.1f} TFLOP/s) solve {st['ms_solve']:.1f} ms residual {resid:.2e} "
          f"device GB {st['bytes_device_total']/1e9:.1f} fail_col {st['fail_col']}", flush=True)
    be.close()

g2 = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g3 = int(sys.argv[2]) if len(sys.argv) > 2 else 48
m = spde.grid_mesh_2d(g2, g2, jitter=0.25, seed=0)
run(f"2-D {g2}x{g2}", m, spde.matern_precision(m, 0, 0.2))
if g3 > 0:
    m3 = spde.grid_mesh_3d(g3, g3, g3)
    run(f"3-D {g3}^3", m3, spde.matern_precision(m3, 0, 0.5), nrhs=16)
