#!/usr/bin/env python3
"""A/B of environment settings on the factorisation of an N^3-node 3-D SPDE (AB_N in the environment, default 80): ms per
refactorisation (best of 3) and the log-determinant, each setting in a child process.    python3 tools/ab_3d.py "" "GMRFX_SYRK_XCD=0" ..."""
import json, os, subprocess, sys
ROOT="/root/repo"
CHILD = r'''
import sys, os, json
sys.path.insert(0, "/root/repo/gaussianmarkovrandomfields.jl_amd")
import numpy as np, torch
import gmrfx
from gmrfx import spde
N=int(os.environ.get("AB_N","80"))
m3 = spde.grid_mesh_3d(N,N,N); Q = spde.matern_precision(m3, 0, 0.4)
n=Q.shape[0]
dev = torch.device("cuda", 0)
be = gmrfx.MI355XBackend(Q, coords=m3.points, device=0, factorize=False)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
t=[]
for k in range(4):
    be.refactorize_dev(d_nz.data_ptr()); t.append(be.stats()["ms_factor"])
print(json.dumps({"ms": float(np.min(t[1:])), "ld": be.compute_logdet()}))
'''
for setting in sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1); env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1]); print(f"{setting or '(defaults)':44s} factor {d['ms']:.1f} ms  logdet {d['ld']:.10e}", flush=True)
    except Exception as e:
        print(setting, "failed", r.stderr[-300:], flush=True)
