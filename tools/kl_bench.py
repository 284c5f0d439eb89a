#!/usr/bin/env python3
"""Timing of the KL (Vecchia) sparse approximate Cholesky batch on one MI355X:
n random points in the unit square, Matern-3/2 covariance (dense, device resident), lower-triangular radius pattern.
usage: tools/kl_bench.py [n] [mean nnz per column]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from gmrfx import klchol
import orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
target = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rng = np.random.default_rng(0)
X = rng.random((n, 2))
rad = np.sqrt(2 * target / (np.pi * n))          # about `target` earlier-or-later neighbours within the radius / 2 per column
P = klchol.radius_pattern(X, rad)
cnt = np.diff(P.indptr)
print(f"n={n} nnz(L)={P.nnz} rows per column: mean {cnt.mean():.1f} max {cnt.max()}", flush=True)
dX = torch.from_numpy(X).cuda()
d = torch.cdist(dX, dX)
K = ((1 + np.sqrt(3) * d / 0.3) * torch.exp(-np.sqrt(3) * d / 0.3)).T.contiguous()   # symmetric; column-major == row-major
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    L = klchol.sparse_approximate_cholesky_inplace(None, P, theta_device_ptr=K.data_ptr())
    t1 = time.perf_counter()
print(f"gmrfx_kl_cholesky (Theta resident, incl. host task set-up and nzval copy back): {1e3*(t1-t0):.1f} ms "
      f"= {n/(t1-t0):.3g} columns/s; gathered {8*(cnt.astype(float)**2).sum()/1e9:.2f} GB, {(cnt.astype(float)**3).sum()/3/1e9:.2f} GFLOP", flush=True)
# CPU restatement on a sample of columns, checked against the GPU result
m = min(n, 1500)
Kh = K[:, :].cpu().numpy()
sub = P[:, :].copy()
t0 = time.perf_counter()
Lo = orc.kl_cholesky_inplace(Kh, P[:, :m].tocsc().__class__((P.data[:P.indptr[m]], P.indices[:P.indptr[m]], P.indptr[:m + 1]), shape=(n, m)))
t1 = time.perf_counter()
err = abs(L[:, :m] - Lo).max() / abs(Lo).max()
print(f"oracle (scipy, 1 thread) on the first {m} columns: {1e3*(t1-t0):.0f} ms = {m/(t1-t0):.3g} columns/s; max rel diff {err:.2e}")
