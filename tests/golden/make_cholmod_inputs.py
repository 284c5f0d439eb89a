#!/usr/bin/env python3
"""Writes the INPUTS of the CHOLMOD parity cases (tests/test_cholmod_parity.py) as the raw files bench/cholmod_baseline.jl
reads: deterministic matrices of the reference's own tests / models, the permutation THIS backend's symbolic analysis picks
(host code only: a symbolic-only handle needs no GPU), and seeded right-hand sides. Run from the repository root:

    python tests/golden/make_cholmod_inputs.py
    for c in tests/golden/cholmod_inputs/*; do julia bench/cholmod_baseline.jl $c tests/golden/cholmod_outputs/$(basename $c); done

The second line needs Julia (absent from the authoring image and from the GPU boxes so far): whoever has one commits the
outputs, and the parity test stops skipping.

    python tests/golden/make_cholmod_inputs.py --grid 1000 [--nrhs 64] [--out gpurun_out/cholmod_in_1000]

writes the inputs of the BENCH-sized case instead (BASELINE cfg 2: the 1000 x 1000-node Matern precision of bench.py, the same
permutation, the same seeded right-hand sides) -- ~0.7 GB, under gpurun_out/ (scratch, not committed). One run of
`julia -t auto bench/cholmod_baseline.jl gpurun_out/cholmod_in_1000 gpurun_out/cholmod_out_1000` anywhere then yields both the
only admissible CPU baseline of the headline (its JSON line: s_refactorize, s_solve, dof_per_s) and bench-sized parity vectors
(scalars.bin: logdet; the first 4096 rows of F \ B and F.UP \ z)."""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import gmrfx  # noqa: E402
from gmrfx import spde  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cholmod_inputs")
NRHS = 4


def cases():
    # test/workspace/test_backend_ordering.jl:9-17 (N = 145: grid Laplacian + dense border)
    nx = 12
    A1 = sp.diags([-np.ones(nx - 1), 2.0 * np.ones(nx), -np.ones(nx - 1)], [-1, 0, 1])
    Qg = sp.kron(sp.identity(nx), A1) + sp.kron(A1, sp.identity(nx)) + 0.1 * sp.identity(nx * nx)
    h = 0.01 * np.ones((nx * nx, 1))
    yield "ordering_145", sp.bmat([[Qg, sp.csr_matrix(h)], [sp.csr_matrix(h.T), sp.csr_matrix(np.array([[2.0]]))]], format="csc"), None
    # src/latent_models/ar.jl:135-148, rho = 0.9, tau = 1, n = 2000
    n, rho = 2000, 0.9
    d = np.full(n, 1 + rho * rho); d[0] = d[-1] = 1.0
    yield "ar1_2000", sp.diags([-rho * np.ones(n - 1), d, -rho * np.ones(n - 1)], [-1, 0, 1], format="csc"), None
    # BASELINE cfg 1 / cfg 2 shapes at CPU-oracle size: 2-D Matern SPDE precisions on jittered P1 meshes
    m = spde.grid_mesh_2d(33, 33, jitter=0.25, seed=0)
    yield "matern2d_33_a2", spde.matern_precision(m, smoothness=0, range_=0.3), m.points
    m = spde.grid_mesh_2d(21, 21, jitter=0.0, seed=0)
    yield "matern2d_21_a3", spde.matern_precision(m, smoothness=1, range_=0.3), m.points


def write_case(d, Q, perm, B):
    n = Q.shape[0]
    os.makedirs(d, exist_ok=True)
    np.array([n, Q.nnz, B.shape[1]], dtype=np.int64).tofile(os.path.join(d, "meta.bin"))
    (Q.indptr.astype(np.int64) + 1).tofile(os.path.join(d, "colptr.bin"))
    (Q.indices.astype(np.int64) + 1).tofile(os.path.join(d, "rowval.bin"))
    Q.data.astype(np.float64).tofile(os.path.join(d, "nzval.bin"))
    (np.asarray(perm, dtype=np.int64) + 1).tofile(os.path.join(d, "perm.bin"))
    np.asfortranarray(B).T.copy().tofile(os.path.join(d, "B.bin"))          # column-major n x nrhs


def bench_sized(grid: int, nrhs: int, out: str):
    """the workload of bench.py (same mesh, range, permutation and right-hand sides: torch.randn((nrhs, n), seed 1))"""
    import torch
    mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
    Q = sp.csc_matrix(spde.matern_precision(mesh, smoothness=0, range_=0.2)); Q.sort_indices()
    n = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True)
    perm = be.ordering_permutation()
    B = torch.randn((nrhs, n), generator=torch.Generator(device="cpu").manual_seed(1), dtype=torch.float64).numpy().T
    write_case(out, Q, perm, B)
    print(f"cfg2 {grid}x{grid}: n = {n}, nnz = {Q.nnz}, nrhs = {nrhs} -> {out}")


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=0, help="write the bench-sized case (grid x grid nodes) instead of the committed small cases")
    ap.add_argument("--nrhs", type=int, default=64)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.grid > 0:
        bench_sized(a.grid, a.nrhs, a.out or os.path.join(ROOT, "gpurun_out", f"cholmod_in_{a.grid}"))
        return
    for name, Q, coords in cases():
        Q = sp.csc_matrix(Q); Q.sort_indices()
        n = Q.shape[0]
        be = gmrfx.MI355XBackend(Q, coords=coords, symbolic_only=True)
        perm = be.ordering_permutation()
        B = np.random.default_rng(7).standard_normal((n, NRHS))
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        np.array([n, Q.nnz, NRHS], dtype=np.int64).tofile(os.path.join(d, "meta.bin"))
        (Q.indptr.astype(np.int64) + 1).tofile(os.path.join(d, "colptr.bin"))
        (Q.indices.astype(np.int64) + 1).tofile(os.path.join(d, "rowval.bin"))
        Q.data.astype(np.float64).tofile(os.path.join(d, "nzval.bin"))
        (np.asarray(perm, dtype=np.int64) + 1).tofile(os.path.join(d, "perm.bin"))
        np.asfortranarray(B).T.copy().tofile(os.path.join(d, "B.bin"))          # column-major n x nrhs
        print(name, "n =", n, "nnz =", Q.nnz)


if __name__ == "__main__":
    main()
