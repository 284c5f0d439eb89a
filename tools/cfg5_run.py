#!/usr/bin/env python3
"""BASELINE.json config 5's operator on ONE MI355X at the largest size that fits: AR(1) (rho = 0.9) x 2-D Matern (alpha = 2)
joint precision kron(Q_t, Q_s) + a diagonal likelihood term (a posterior precision: block tridiagonal in time), factored
by nested dissection of the space-time graph; refactorise + 64-RHS solve + logdet, inputs resident in HBM.
usage: cfg5_run.py [T] [G] (T time steps x G*G spatial nodes) [--max-gb G]. The full config (512 x 250 000 = 1.28e8
unknowns) needs a ~0.5 TB top front and is out of reach of one node."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np
import torch    # before libgmrfx.so (one HIP runtime per process)

ap = argparse.ArgumentParser()
ap.add_argument("T", type=int, nargs="?", default=128)
ap.add_argument("G", type=int, nargs="?", default=125)
ap.add_argument("--nrhs", type=int, default=64)
ap.add_argument("--max-gb", type=float, default=250.0)
args = ap.parse_args()
import gmrfx
from gmrfx import spde
t0 = time.time()
m = spde.grid_mesh_2d(args.G, args.G, jitter=0.25, seed=1)
Qt = spde.ar1_precision(args.T, 0.9, 1.0)
Qs = spde.matern_precision(m, 0, 0.2)
ns = Qs.shape[0]
Q = gmrfx.spacetime_precision(Qt, Qs, obs_diag=np.random.default_rng(0).uniform(0.5, 2.0, args.T * ns))
n = Q.shape[0]
coords = gmrfx.spacetime_coords(m.points, args.T)
print(f"generated n={n} nnz(Q)={Q.nnz} in {time.time()-t0:.0f}s", flush=True)
t0 = time.time()
sym = gmrfx.MI355XBackend(Q, coords=coords, symbolic_only=True)
st = sym.stats(); perm = sym.ordering_permutation(); sym.close()
need = (st["bytes_factor"] + st["bytes_cb_arena"] + 8.0 * Q.nnz + 3 * 8.0 * 64 * n + 8.0 * 64 * (st["sum_rows"] - n)) / 1e9
print(f"symbolic {time.time()-t0:.0f}s: nnz(L)={st['nnz_l']:.3e} flops={st['factor_flops']:.3e} max front {st['max_cols']} cols, predicted {need:.0f} GB", flush=True)
if need > args.max_gb:
    print(json.dumps({"workload": f"cfg5 {args.T} x {args.G}^2", "n": n, "status": "refused", "predicted_gb": need})); sys.exit(0)
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
be = gmrfx.MI355XBackend(Q, ordering=perm, device=0, factorize=False)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
Bh = torch.randn((args.nrhs, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64)
d_B = Bh.to(dev); d_X = torch.empty_like(d_B); torch.cuda.synchronize()
tf, ts = [], []
for k in range(2):
    be.refactorize_dev(d_nz.data_ptr()); be.solve_dev(d_B.data_ptr(), n, args.nrhs, d_X.data_ptr(), n)
    s = be.stats(); tf.append(s["ms_factor"]); ts.append(s["ms_solve"])
X = d_X.cpu().numpy().T
resid = float(np.linalg.norm(Q @ X - Bh.numpy().T) / np.linalg.norm(Bh.numpy()))
s = be.stats(); mf, ms = min(tf), min(ts)
print(json.dumps({"workload": f"cfg5 operator: AR(1) x 2-D Matern posterior precision, {args.T} time steps x {args.G}^2 nodes, space-time nested dissection, ONE MI355X",
                  "n": n, "nnz_Q": int(Q.nnz), "nnz_L": int(s["nnz_l"]), "factor_flops": s["factor_flops"], "max_front_cols": int(s["max_cols"]),
                  "ms_factor": mf, "ms_solve": ms, "factor_TFLOPs": s["factor_flops"] / mf / 1e9, "DoF_per_s": n / ((mf + ms) * 1e-3),
                  "rel_residual": resid, "logdet": be.compute_logdet(), "fail_col": s["fail_col"], "device_GB": s["bytes_device_total"] / 1e9}))
