#!/usr/bin/env python3
"""One mode of tools/fused_step.py for the profiler: `separate` or `pipelined`, 3 warm-up + 20 steps of cfg 2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde
mode = sys.argv[1]
mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, 0, 0.2)
n = Q.shape[0]
be = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
dev = torch.device("cuda", 0)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
d_B = torch.randn((64, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
d_X = torch.empty_like(d_B)
torch.cuda.synchronize()
for _ in range(23):
    if mode == "separate":
        be.refactorize_dev(d_nz.data_ptr()); be.solve_dev(d_B.data_ptr(), n, 64, d_X.data_ptr(), n)
    else:
        be.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, 64, d_X.data_ptr(), n)
torch.cuda.synchronize()
be.close()
