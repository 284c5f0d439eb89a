#!/usr/bin/env python3
"""Sweep tasks A/B: forward / backward sweep time at cfg 2 for different LDS row caps of the task local vector
(GMRFX_SWEEP_TASK_ROWS: 0 = no tasks, pure level schedule; default 288)."""
import os, subprocess, sys
code = r'''
import os, sys
sys.path.insert(0, "gaussianmarkovrandomfields.jl_amd")
import numpy as np, torch, gmrfx
from gmrfx import spde
mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, 0, 0.2); n = Q.shape[0]
be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
dev = torch.device("cuda", 0)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
d_B = torch.randn((64, n), dtype=torch.float64).to(dev); d_X = torch.empty_like(d_B); torch.cuda.synchronize()
be.refactorize_dev(d_nz.data_ptr())
for k in range(4):
    be.solve_dev(d_B.data_ptr(), n, 64, d_X.data_ptr(), n)
    s = be.stats()
print(os.environ.get("GMRFX_TASK_DBG", "0"), os.environ.get("GMRFX_SWEEP_TASK_ROWS", "-"), "fwd %.3f bwd %.3f" % (s["ms_solve_fwd"], s["ms_solve_bwd"]), flush=True)
'''
for dbg, rows in (("0", None), ("0", "0"), ("0", "144"), ("0", "208")):
    env = dict(os.environ, GMRFX_TASK_DBG=dbg)
    if rows is not None:
        env["GMRFX_SWEEP_TASK_ROWS"] = rows
    subprocess.run([sys.executable, "-c", code], env=env)
