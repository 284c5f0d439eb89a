#!/usr/bin/env python3
"""A/B of environment settings on the cfg-3 selected inversion (10^6-node 2-D SPDE; AB_3D=N in the environment: an N^3-node 3-D one):
ms per call, each setting in a child process.
    python3 tools/ab_selinv.py "" "GMRFX_INV_CAP=128" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, os.path.join(%r, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde
n3 = int(os.environ.get("AB_3D", "0"))          # AB_3D=N: an N^3-node 3-D SPDE (cfg 4's family) instead of cfg 3's 2-D one
mesh = spde.grid_mesh_3d(n3, n3, n3) if n3 else spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, 0, 0.4 if n3 else 0.2)
be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0)
t = []
for k in range(6):
    be.selinv_compute_dev()
    t.append(be.stats()["ms_selinv"])
d = be.get_selinv_diag()
print(json.dumps({"ms": float(np.median(t[1:])), "diag_sum": float(d.sum()), "diag_min": float(d.min()), "diag_max": float(d.max())}))
''' % ROOT
for setting in sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{setting or '(defaults)':40s} selected inversion {d['ms']:.3f} ms | sum diag {d['diag_sum']:.10e} min {d['diag_min']:.6f} max {d['diag_max']:.6f}", flush=True)
    except Exception as e:
        print(f"{setting}: failed ({e!r}); stderr tail: {r.stderr[-300:]}", flush=True)
