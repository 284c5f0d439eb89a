"""CPU: the oracle against (a) the committed golden fixtures, (b) the dense identities the
reference's own tests use (test/workspace/test_gmrf_workspace.jl:26-83, test/test_gmrf.jl:64-76),
(c) scipy SuperLU as an independent sparse solver."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import orc
from gmrfx import spde

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load(path):
    g = np.load(path)
    n = int(g["n"])
    Q = sp.csc_matrix((g["nzval"], g["rowval"], g["colptr"]), shape=(n, n))
    return g, Q


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_matches_golden(path):
    g, Q = load(path)
    F = orc.OracleFactor(Q, g["perm"])
    assert np.array_equal(np.diff(F.L().indptr), g["L_colcount"])          # integer work: exact
    assert np.allclose(F.L().diagonal(), g["L_diag"], rtol=1e-11)
    assert np.allclose(F.solve(g["B"]), g["X"], rtol=1e-10, atol=1e-13)
    assert np.allclose(F.backward_solve(g["Z"]), g["Xb"], rtol=1e-10, atol=1e-13)
    assert np.isclose(F.logdet(), float(g["logdet"]), rtol=1e-12)
    assert np.allclose(F.selinv_diag(), g["selinv_diag"], rtol=1e-9)
    Z = F.selinv()
    cols = np.repeat(np.arange(Q.shape[0]), np.diff(Q.indptr))
    assert np.allclose(np.asarray(Z[Q.indices, cols]).ravel(), g["Qinv_on_pattern"], rtol=1e-8, atol=1e-13)


@pytest.mark.parametrize("n", [20, 100, 400])
def test_oracle_dense_identities_reference_fixture(n):
    """`Q = S S' + n I` (test_gmrf_workspace.jl:8-13) at the reference's sizes."""
    Q = spde.random_spd_precision(n, 0.3 if n < 400 else 0.02)
    D = Q.toarray()
    Qi = np.linalg.inv(D)
    F = orc.OracleFactor(Q)                      # identity permutation
    b = np.random.default_rng(n).standard_normal(n)
    assert np.allclose(F.solve(b), np.linalg.solve(D, b), rtol=1e-10)
    assert np.isclose(F.logdet(), np.linalg.slogdet(D)[1], rtol=1e-10)
    assert np.allclose(F.selinv_diag(), np.diag(Qi), rtol=1e-8)
    Z = F.selinv().tocoo()
    assert np.allclose(Z.data, Qi[Z.row, Z.col], rtol=1e-6, atol=1e-14)
    assert np.isclose((F.selinv().multiply(Q)).sum(), n, rtol=1e-8)      # tr(Q^-1 Q) = n
    # uplo handling: only the selected triangle defines Q
    Qu = sp.triu(Q, format="csc")
    assert np.isclose(orc.OracleFactor(Qu, uplo="U").logdet(), F.logdet(), rtol=1e-13)
    assert np.isclose(orc.OracleFactor(sp.tril(Q, format="csc"), uplo="L").logdet(), F.logdet(), rtol=1e-13)


def test_oracle_vs_superlu_medium():
    m = spde.grid_mesh_2d(60, 60, jitter=0.25)
    Q = spde.matern_precision(m, 0, 0.2)
    F = orc.OracleFactor(Q, np.random.default_rng(0).permutation(m.n))
    lu = spla.splu(Q, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    B = np.random.default_rng(1).standard_normal((m.n, 4))
    assert np.allclose(F.solve(B), lu.solve(B), rtol=1e-9, atol=1e-12)
    assert np.isclose(F.logdet(), np.log(np.abs(lu.U.diagonal())).sum() + np.log(np.abs(lu.L.diagonal())).sum(), rtol=1e-11)


def test_backward_solve_is_a_sampler():
    """cov(P' L^-T z) = Q^-1 in ORIGINAL ordering (test_gmrf_workspace.jl:85-100)."""
    Q = spde.random_spd_precision(20)
    F = orc.OracleFactor(Q, np.random.default_rng(3).permutation(20))
    S = F.backward_solve(np.random.default_rng(123).standard_normal((20, 50000)))
    assert np.allclose((S * S).mean(axis=1), np.diag(np.linalg.inv(Q.toarray())), rtol=0.1)


def test_indefinite_is_flagged_not_fatal():
    Q = spde.random_spd_precision(15).tolil()
    Q.setdiag(-1.0)
    F = orc.OracleFactor(sp.csc_matrix(Q))
    assert F.fail_col >= 0


def test_matern_generator_matches_reference_formulas():
    """Appendix A of SURVEY.md: variance ~ 1 in the interior, pattern sizes, explicit zeros kept."""
    m = spde.grid_mesh_2d(41, 41)
    Q = spde.matern_precision(m, smoothness=0, range_=0.3)       # nu = 1, alpha = 2: 19-point pattern
    assert np.diff(Q.indptr).max() == 19 and (Q.data == 0).sum() > 0
    var = orc.OracleFactor(Q).selinv_diag().reshape(41, 41)
    assert abs(var[20, 20] - 1.0) < 0.1                           # sigma^2 = 1 away from the boundary
    assert np.diff(spde.matern_precision(m, 1, 0.3).indptr).max() == 37   # alpha = 3
    with pytest.raises(ValueError):
        spde.smoothness_to_nu(-1, 2)
    # 3-D nu=1 (alpha = 5/2) is not expressible, as in the reference (Integer(alpha) throws)
    import math
    assert spde.smoothness_to_nu(0, 3) == 0.5 and math.isclose(spde.range_to_kappa(2.0, 1.0), math.sqrt(8) / 2)


def test_oracle_sqmahal_and_logpdf_dense_identity():
    # dot(r, Q r) and the unconstrained logpdf of workspace_gmrf.jl:288-292 against the dense formulas
    Q = sp.csc_matrix(spde.random_spd_precision(60, 0.1))
    n = Q.shape[0]
    rng = np.random.default_rng(0)
    z, mu = rng.standard_normal(n), rng.standard_normal(n)
    r = z - mu
    assert abs(orc.sqmahal(Q, z, mu) - r @ Q.toarray() @ r) < 1e-10
    assert abs(orc.sqmahal(sp.tril(Q).tocsc(), z, mu, uplo="L") - r @ Q.toarray() @ r) < 1e-10
    assert abs(orc.sqmahal(sp.triu(Q).tocsc(), z, None) - z @ Q.toarray() @ z) < 1e-10
    F = orc.OracleFactor(Q)
    from scipy.stats import multivariate_normal
    want = multivariate_normal(mean=mu, cov=np.linalg.inv(Q.toarray())).logpdf(z)
    assert abs(orc.logpdf(F, Q, z, mu) - want) < 1e-8 * max(1.0, abs(want))


def test_oracle_selinv_contractions_dense_identity():
    # chain-like pattern with full fill inside the band: Sigma on the factor pattern is exact, so the restated
    # _row_diag_AΣAt / selinv_dot agree with dense algebra whenever the pairs stay inside the pattern
    Q = sp.csc_matrix(spde.matern_precision(spde.grid_mesh_2d(9, 9), 0, 0.4))
    n = Q.shape[0]
    F = orc.OracleFactor(Q)
    Sigma = np.linalg.inv(Q.toarray())
    assert abs(orc.selinv_dot(F, Q) - n) < 1e-9 * n
    A = sp.csr_matrix(abs(Q[:40]))          # rows of Q: pairs within a row are within distance 2 hops x 2 -> check via oracle pattern
    Sig = F.selinv().toarray()
    want = np.einsum("ij,jk,ik->i", A.toarray(), Sig, A.toarray())      # Sigma = 0 outside the pattern, as the reference documents
    assert np.abs(orc.row_diag_ASigmaAt(F, A) - want).max() < 1e-10 * np.abs(want).max()
    pat = (F.selinv() != 0).toarray()
    assert np.abs((Sig - Sigma)[pat]).max() < 1e-9


def test_oracle_kl_cholesky_reference_identities():
    # the known answers of the reference's own tests (test/kl_cholesky/test_sparse_cholesky.jl:14-40): diagonal
    # Theta with the full lower pattern gives L L' = inv(Theta) (atol 1e-4 because of the 1e-6 I), L lower
    # triangular with a positive diagonal; and, for ANY SPD Theta, the full pattern makes L L' exactly
    # (Theta + 1e-6 I)^-1 -- the KL-optimal factor of a complete pattern is the exact inverse factor
    Theta = np.diag([1.0, 2.0, 3.0, 4.0])
    full = sp.csc_matrix(np.tril(np.ones((4, 4))))
    L = orc.kl_cholesky_inplace(Theta, full)
    assert abs(sp.triu(L, 1)).sum() == 0
    assert np.abs((L @ L.T).toarray() - np.linalg.inv(Theta)).max() < 1e-4
    rng = np.random.default_rng(42)
    A = rng.random((6, 6)); Theta = (A + A.T) / 2 + 6 * np.eye(6)
    L = orc.kl_cholesky_inplace(Theta, sp.csc_matrix(np.tril(np.ones((6, 6)))))
    assert (L.diagonal() > 0).all()
    assert np.abs((L @ L.T).toarray() - np.linalg.inv(Theta + 1e-6 * np.eye(6))).max() < 1e-12
    # one supernode holding every column = the same complete pattern (regularisation 1e-8 there)
    Ls = orc.kl_cholesky_supernodal(Theta, [list(range(6))], [list(range(5, -1, -1))])
    assert np.abs((Ls @ Ls.T).toarray() - np.linalg.inv(Theta + 1e-8 * np.eye(6))).max() < 1e-12
    # supernodes whose members have exactly the supernode's pattern agree with the column-wise routine
    # (test_sparse_cholesky.jl:82-108 compares the two variants through their approximation error)
    X = np.stack(np.meshgrid(np.arange(0, 1.01, 0.3), np.arange(0, 1.01, 0.3)), -1).reshape(-1, 2)
    K = np.exp(-((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / 0.4)
    n = len(X)
    cols = [[k] for k in range(n)]
    rows = [sorted([i for i in range(k, n) if np.linalg.norm(X[i] - X[k]) < 0.7], reverse=True) for k in range(n)]
    pat = sp.csc_matrix((np.ones(sum(map(len, rows))), (np.concatenate(rows), np.repeat(np.arange(n), list(map(len, rows))))), shape=(n, n))
    La = orc.kl_cholesky_inplace(K, pat, reg=1e-8)
    Lb = orc.kl_cholesky_supernodal(K, cols, rows, reg=1e-8)
    assert abs(La - Lb).max() < 1e-12 * abs(La).max()


def test_supernodal_cpu_baseline_matches_simplicial_oracle():
    """oracle/supernodal_cpu.c (bench.py's multi-threaded cpu_baseline: supernodal multifrontal LL' on OpenBLAS
    kernels, on the symbolic structure libgmrfx's host analysis exports) against the simplicial oracle: factor
    values, logdet, multi-RHS solve, backward solve -- and through it the exported supernode / relative-index /
    scatter structures once more."""
    import gmrfx
    import sncpu
    from gmrfx import spde
    for mesh, rng_ in ((spde.grid_mesh_2d(70, 55, jitter=0.25, seed=2), 0.25), (spde.grid_mesh_3d(11, 9, 10), 0.5)):
        Q = sp.csc_matrix(spde.matern_precision(mesh, 0, rng_))
        n = Q.shape[0]
        be = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True)
        perm = be.ordering_permutation()
        sn = sncpu.SupernodalCPU(be.symbolic(), perm, n, nthreads=4)
        assert sn.factorize(Q.data) == -1
        F = orc.OracleFactor(Q, perm)
        assert abs(sn.logdet() - F.logdet()) < 1e-11 * abs(F.logdet())
        B = np.random.default_rng(0).standard_normal((n, 7))
        assert np.abs(sn.solve(B) - F.solve(B)).max() < 1e-10 * np.abs(F.solve(B)).max()
        assert np.abs(sn.solve(B[:, 0], mode=1) - F.backward_solve(B[:, 0])).max() < 1e-10
        bad = Q.copy(); bad.data = -bad.data
        assert sn.factorize(bad.data) >= 0


def test_oracle_on_the_references_deterministic_inputs():
    """The oracle itself on the inputs the reference's tests fully specify (no RNG in Q): the 145 x 145 grid Laplacian +
    dense border of test/workspace/test_backend_ordering.jl:9-17 under `perm = N:-1:1`, the natural order and a random
    one (answers must not depend on the ordering, :25-31: solve / logdet rtol 1e-10, selinv diagonal 1e-8), and the AR(1)
    precision of src/latent_models/ar.jl:135-148 against its closed forms det = tau^n (1 - rho^2),
    Sigma_ij = rho^|i-j| / (tau (1 - rho^2))."""
    from test_reference_inputs import backend_ordering_matrix
    from gmrfx import spde
    _, Q = backend_ordering_matrix()
    N = Q.shape[0]
    Qd = Q.toarray()
    rhs = np.random.default_rng(0).standard_normal(N)
    x_ref, ld_ref, d_ref = np.linalg.solve(Qd, rhs), np.linalg.slogdet(Qd)[1], np.diag(np.linalg.inv(Qd))
    for perm in (np.arange(N - 1, -1, -1), np.arange(N), np.random.default_rng(1).permutation(N)):
        F = orc.OracleFactor(Q, perm)
        assert np.abs(F.solve(rhs) - x_ref).max() < 1e-10 * np.abs(x_ref).max()
        assert abs(F.logdet() - ld_ref) < 1e-10 * abs(ld_ref)
        assert np.abs(F.selinv_diag() - d_ref).max() < 1e-8 * d_ref.max()
        L = F.L().toarray()
        assert np.abs(L @ L.T - Qd[np.ix_(perm, perm)]).max() < 1e-12 * np.abs(Qd).max()
    n, rho, tau = 400, 0.9, 2.5
    Qa = sp.csc_matrix(spde.ar1_precision(n, rho, tau))
    F = orc.OracleFactor(Qa, np.arange(n))
    s2 = 1.0 / (tau * (1.0 - rho * rho))
    assert abs(F.logdet() - (n * np.log(tau) + np.log1p(-rho * rho))) < 1e-11 * n
    assert np.abs(F.selinv_diag() - s2).max() < 1e-9 * s2
    e = np.zeros(n); e[137] = 1.0
    assert np.allclose(F.solve(e), s2 * rho ** np.abs(np.arange(n) - 137), rtol=1e-9, atol=1e-14)
