#!/usr/bin/env python3
"""A/B of environment settings on wide passes at cfg 2 / cfg 3: 256 samples (backward solve) and a 256-column solve, ms per call,
each setting in a child process.    python3 tools/ab_rand.py "" "GMRFX_BWD_FRONT=0" ..."""
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
be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0)
d_Z = torch.randn((256, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
d_X = torch.empty_like(d_Z)
out = {}
for name, call, key, k in (("rand256", be.backward_solve_dev, "ms_backward_solve", 256), ("solve256", be.solve_dev, "ms_solve", 256),
                           ("rand128", be.backward_solve_dev, "ms_backward_solve", 128)):
    t = []
    for rep in range(8):
        call(d_Z.data_ptr(), n, k, d_X.data_ptr(), n)
        t.append(be.stats()[key])
    out[name] = float(np.median(t[2:]))
out["sum"] = float(d_X[:128].sum().item())
print(json.dumps(out))
''' % ROOT
for setting in sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{setting or '(defaults)':32s} 256 samples {d['rand256']:.3f} ms | 256-column solve {d['solve256']:.3f} ms | 128 samples {d['rand128']:.3f} ms | checksum {d['sum']:.10e}", flush=True)
    except Exception as e:
        print(f"{setting}: failed ({e!r}); stderr tail: {r.stderr[-300:]}", flush=True)
