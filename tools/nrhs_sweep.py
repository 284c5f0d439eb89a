#!/usr/bin/env python3
"""Solve / backward-solve time at cfg 2 as a function of the number of right-hand sides (device-resident panels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, smoothness=0, range_=0.2)
n = Q.shape[0]
be = gmrfx.MI355XBackend(Q, coords=mesh.points)
for nr in [int(x) for x in os.environ.get("NRHS_LIST", "1,2,4,8,16,17,24,32,33,48,64,128,256").split(",")]:
    B = torch.randn((nr, n), dtype=torch.float64, device="cuda"); X = torch.empty_like(B)
    torch.cuda.synchronize()
    for rep in range(3):
        be.solve_dev(B.data_ptr(), n, nr, X.data_ptr(), n)
    s = be.stats()
    for rep in range(2):
        be.backward_solve_dev(B.data_ptr(), n, nr, X.data_ptr(), n)
    s2 = be.stats()
    print(f"nrhs {nr:4d}: solve {s['ms_solve']:7.3f} ms (fwd {s['ms_solve_fwd']:.3f} bwd {s['ms_solve_bwd']:.3f} perm {s['ms_solve_perm']:.3f})  backward_solve {s2['ms_backward_solve']:7.3f} ms", flush=True)
