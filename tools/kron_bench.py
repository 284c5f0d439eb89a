#!/usr/bin/env python3
"""cfg 5 of SURVEY 8d at the PRIOR level: AR(1) rho = 0.9 over 512 time steps (x) 500 x 500-node 2-D Matern alpha = 2
= 1.28e8 unknowns, answered from the two factor-scale factorisations (gmrfx.KroneckerWorkspace): logdet, one
sample, marginal variances. usage: tools/kron_bench.py [grid] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np
import gmrfx
from gmrfx import spde

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 500
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
Qs = spde.matern_precision(mesh, smoothness=0, range_=0.2)
Qt = spde.ar1_precision(T, 0.9)
t0 = time.perf_counter()
kw = gmrfx.KroneckerWorkspace(Qt, Qs, kw2={"coords": mesh.points})
t1 = time.perf_counter()
N = kw.dimension()
print(f"N = {T} x {Qs.shape[0]} = {N:.3e}; workspaces (symbolic + first numeric of both factors): {t1-t0:.2f} s", flush=True)
t0 = time.perf_counter(); ld = kw.logdet(); t1 = time.perf_counter()
print(f"logdet(Q_t (x) Q_s) = {ld:.6e}: {1e3*(t1-t0):.1f} ms", flush=True)
z = np.random.default_rng(0).standard_normal(N)
t0 = time.perf_counter(); x = kw.backward_solve(z); t1 = time.perf_counter()
print(f"one sample of all {N:.3e} unknowns (host arrays in and out): {t1-t0:.2f} s; "
      f"GPU time of the spatial sweep ({T} RHS): {kw.ws2.stats()['ms_backward_solve']:.1f} ms", flush=True)
import torch
zd = torch.from_numpy(z).cuda()
xd = kw.backward_solve_dev(zd); torch.cuda.synchronize()           # first call: the dense operator of the small factor
t0 = time.perf_counter(); xd = kw.backward_solve_dev(zd); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"one sample, z and x resident in HBM (backward_solve_dev): {1e3*(t1-t0):.1f} ms wall; "
      f"max |x_dev - x_host| / max |x| = {float((xd.cpu() - torch.from_numpy(x)).abs().max() / np.abs(x).max()):.1e}", flush=True)
t0 = time.perf_counter(); sd = kw.solve_dev(zd); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"one solve of all unknowns, resident (solve_dev): {1e3*(t1-t0):.1f} ms wall", flush=True)
t0 = time.perf_counter(); v = kw.selinv_diag(); t1 = time.perf_counter()
print(f"marginal variances: {t1-t0:.2f} s (min {v.min():.3e}, max {v.max():.3e}); sample variance / mean marginal variance = {x.var()/v.mean():.3f}")
