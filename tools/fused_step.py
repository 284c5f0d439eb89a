#!/usr/bin/env python3
"""A/B of the pipelined factor + solve call (gmrfx_refactorize_solve_dev) against the separate calls on the bench workload
(cfg 2: 1000 x 1000-node mesh, 64 right-hand sides, inputs resident in HBM): wall time per step, device phases, equality of X.

    python3 tools/fused_step.py [grid=1000] [nrhs=64] [steps=20]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mesh = spde.grid_mesh_2d(G, G, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, 0, 0.2)
n = Q.shape[0]
be = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
dev = torch.device("cuda", 0)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
d_B = torch.randn((nrhs, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
d_X, d_Y = torch.empty_like(d_B), torch.empty_like(d_B)
torch.cuda.synchronize()


def separate():
    be.refactorize_dev(d_nz.data_ptr()); be.solve_dev(d_B.data_ptr(), n, nrhs, d_X.data_ptr(), n)


def fused():
    be.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, nrhs, d_Y.data_ptr(), n)


out = {"workload": f"{G}x{G} mesh, {nrhs} RHS", "n": n}
for name, fn in (("separate", separate), ("pipelined", fused), ("separate_again", separate), ("pipelined_again", fused)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ph = []
    for _ in range(steps):
        fn()
        s = be.stats()
        ph.append((s["ms_factor"], s["ms_solve"], s["ms_solve_fwd"], s["ms_solve_bwd"]))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    ph = np.median(np.asarray(ph), axis=0)
    out[name] = {"ms_per_step": ms, "ms_factor": float(ph[0]), "ms_solve": float(ph[1]), "ms_fwd": float(ph[2]), "ms_bwd": float(ph[3])}
    print(name, {k: round(v, 3) for k, v in out[name].items()}, flush=True)
out["bit_identical"] = bool(torch.equal(d_X, d_Y))
print(json.dumps(out))
