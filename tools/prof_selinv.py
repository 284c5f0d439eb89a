#!/usr/bin/env python3
"""Profile target: refactorise once, then K selected inversions (+ optional 256-sample backward solves) at cfg 2."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, smoothness=0, range_=0.2)
n = Q.shape[0]
be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
dev = torch.device("cuda", 0)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
ts = []
for _ in range(K):
    be.refactorize_dev(d_nz.data_ptr())
    be.selinv_compute_dev()
    ts.append(be.stats()["ms_selinv"])
print("ms_selinv", ts)
