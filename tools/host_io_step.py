"""The pipelined factor + solve step through the HOST entry point (gmrfx_refactorize_solve, what workspace_solve(ws, B::Matrix) calls)
against the device-resident call: wall time per step and the library's device-side phases, for pageable and page-locked arrays.
usage: python tools/host_io_step.py [grid] [nrhs] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np
import torch
import gmrfx
from gmrfx import spde

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, smoothness=0, range_=0.2)
n = Q.shape[0]
be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
dev = torch.device("cuda", 0)
Bh = torch.randn((nrhs, n), generator=torch.Generator(device="cpu").manual_seed(1), dtype=torch.float64)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
d_B = Bh.to(dev); d_X = torch.empty_like(d_B)
for _ in range(3):
    be.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, nrhs, d_X.data_ptr(), n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    be.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, nrhs, d_X.data_ptr(), n)
torch.cuda.synchronize()
print(f"device-resident: {1e3 * (time.perf_counter() - t0) / reps:.2f} ms per step")
ref = d_X.cpu()
nz_pg = np.ascontiguousarray(Q.data)
nz_pin = torch.from_numpy(nz_pg).clone().pin_memory()
for kind in ("pageable", "pinned"):
    if kind == "pageable":
        Bf = np.asfortranarray(Bh.numpy().T); Xf = np.zeros_like(Bf, order="F")
        bp, xp, nzp = Bf.ctypes.data, Xf.ctypes.data, nz_pg.ctypes.data
    else:
        Bp = Bh.clone().pin_memory(); Xp = torch.zeros_like(Bp).pin_memory()
        bp, xp, nzp = Bp.data_ptr(), Xp.data_ptr(), nz_pin.data_ptr()
    be.refactorize_solve_ptr(nzp, bp, n, nrhs, xp, n)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        be.refactorize_solve_ptr(nzp, bp, n, nrhs, xp, n)
        ts.append(1e3 * (time.perf_counter() - t0))
    s = be.stats()
    Xh = torch.from_numpy(Xf.T) if kind == "pageable" else Xp
    print(f"{kind:9s}: {np.median(ts):.2f} ms per step (min {min(ts):.2f}); device phases: factor {s['ms_factor']:.2f}, behind the factor {s['ms_solve']:.2f} "
          f"(fwd left {s['ms_solve_fwd']:.2f}, bwd {s['ms_solve_bwd']:.2f}, transposes {s['ms_solve_perm']:.2f}); equal bits: {bool(torch.equal(Xh, ref))}")
