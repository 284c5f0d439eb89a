#!/usr/bin/env python3
"""A/B of environment switches on the cfg-2 solve (separate call; 64 right-hand sides, AB_NRHS=k in the environment: k): forward / backward sweep ms per setting, each in a child process.
    python3 tools/ab_solve.py "" "GMRFX_BWD_FRONT=0" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, os.path.join(%r, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde
mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, 0, 0.2)
n = Q.shape[0]
dev = torch.device("cuda", 0)
be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
NR = int(os.environ.get("AB_NRHS", "64"))
d_B = torch.randn((NR, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
d_X = torch.empty_like(d_B)
be.refactorize_dev(d_nz.data_ptr())
f, b = [], []
for k in range(12):
    be.solve_dev(d_B.data_ptr(), n, NR, d_X.data_ptr(), n)
    st = be.stats()
    if k >= 2: f.append(st["ms_solve_fwd"]); b.append(st["ms_solve_bwd"])
print(json.dumps({"fwd": float(np.median(f)), "bwd": float(np.median(b))}))
''' % ROOT
for setting in sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{setting or '(defaults)':40s} forward {d['fwd']:.3f} ms | backward {d['bwd']:.3f} ms", flush=True)
    except Exception as e:
        print(f"{setting}: failed ({e!r}); stderr tail: {r.stderr[-300:]}", flush=True)
