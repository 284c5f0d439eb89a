#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/*.npz.

The reference (GaussianMarkovRandomFields.jl) cannot run here (pure Julia; no Julia in the image)
and ships no stored vectors for this path -- its own tests compare against dense
inv/logdet/solve built on the spot (test/workspace/test_gmrf_workspace.jl:26-57). The fixtures
are therefore produced the same way: inputs from the in-repo SPDE generator / the reference's
`Q = S S' + n I` fixture recipe, expected outputs from DENSE numpy linear algebra in float64,
cross-checked inside this script against two independent sparse solvers (scipy SuperLU and the
C oracle). Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
from gmrfx import spde  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, Q, perm, seed):
    Q = sp.csc_matrix(Q)
    Q.sort_indices()
    n = Q.shape[0]
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, 3))
    Zs = rng.standard_normal((n, 2))
    D = Q.toarray()
    Dinv = np.linalg.inv(D)
    X = np.linalg.solve(D, B)
    logdet = np.linalg.slogdet(D)[1]
    P = np.eye(n)[perm]                     # (P Q P') = L L'
    Lp = np.linalg.cholesky(P @ D @ P.T)
    Xb = P.T @ np.linalg.solve(Lp.T, Zs)    # F.UP \ z
    # cross-checks: SuperLU and the C oracle must agree with the dense answers
    lu = spla.splu(Q, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    assert np.allclose(lu.solve(B), X, rtol=1e-10, atol=1e-13)
    assert np.isclose(np.log(np.abs(lu.U.diagonal())).sum() + np.log(np.abs(lu.L.diagonal())).sum(), logdet, rtol=1e-12)
    F = orc.OracleFactor(Q, perm)
    assert np.allclose(F.solve(B), X, rtol=1e-10, atol=1e-13)
    assert np.allclose(F.backward_solve(Zs), Xb, rtol=1e-10, atol=1e-13)
    assert np.isclose(F.logdet(), logdet, rtol=1e-12)
    assert np.allclose(F.selinv_diag(), np.diag(Dinv), rtol=1e-9)
    Lo = F.L().toarray()
    assert np.allclose(Lo, Lp, rtol=1e-9, atol=1e-12)
    colcount = np.diff(F.L().indptr)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), n=n, colptr=Q.indptr.astype(np.int64), rowval=Q.indices.astype(np.int64),
        nzval=Q.data, perm=np.asarray(perm, dtype=np.int64), B=B, X=X, Z=Zs, Xb=Xb, logdet=logdet,
        selinv_diag=np.diag(Dinv).copy(),
        Qinv_on_pattern=Dinv[Q.indices, np.repeat(np.arange(n), np.diff(Q.indptr))].copy(), L_colcount=colcount,
        L_diag=np.diag(Lp).copy())
    print(f"{name}: n={n} nnz={Q.nnz} nnz(L)={colcount.sum()} logdet={logdet:.12g}")


if __name__ == "__main__":
    rng = np.random.default_rng(2024)
    m = spde.grid_mesh_2d(16, 16, jitter=0.25, seed=0)
    make("matern2d_16x16_a2", spde.matern_precision(m, 0, 0.4), rng.permutation(m.n), 1)
    m = spde.grid_mesh_2d(13, 13)
    make("matern2d_13x13_a3_structural_zeros", spde.matern_precision(m, 1, 0.3), rng.permutation(m.n), 2)
    m3 = spde.grid_mesh_3d(6, 6, 6)
    make("matern3d_6_a2", spde.matern_precision(m3, 0, 0.6), rng.permutation(m3.n), 3)
    make("sprand_spd_60", spde.random_spd_precision(60, 0.3, seed=42), rng.permutation(60), 4)
    Qst = sp.kron(spde.ar1_precision(6, 0.9), spde.matern_precision(spde.grid_mesh_2d(7, 7), 0, 0.5)).tocsc()
    make("ar1_kron_matern_6x49", Qst, rng.permutation(Qst.shape[0]), 5)
