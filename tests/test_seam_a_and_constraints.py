"""GPU tests of the reference's call sequences ABOVE the solver seams, driven through libgmrfx.so:

  * seam A -- the LinearSolve cache protocol (`init` / `solve!` / `cache.A = ...` / `cache.b = ...` /
    `deepcopy(cache)`), the GMRF hooks the Pardiso extension overloads, `GMRF(mean, Q, alg)` and
    `GMRF(InformationVector, Q, alg)`: test/test_linearsolve_architecture.jl:12-100, test/test_gmrf.jl:36-76,
    src/arithmetic/condition/gaussian_approximation.jl:41-125, 428-499 (Newton loop, constraint projection)
  * `ordering` keyword forms on answers: test/workspace/test_backend_ordering.jl:33-60
  * the constraint path of seam B: src/workspace/workspace_gmrf.jl:22-56, 260-305 (one n x m multi-RHS solve,
    m x m Cholesky, constrained mean / var / rand / logpdf)

Every expected value is a dense numpy identity, as in the reference's own tests."""
import copy

import numpy as np
import pytest
import scipy.sparse as sp

import gmrfx
from gmrfx import PinDenseColumns, spde
from mirror import GMRFWorkspace, WorkspacePool
from mirror import linsolve as ls
from mirror.workspace_gmrf import ConstraintInfo, WorkspaceGMRF
from test_ordering_kwarg import mmd
from test_reference_inputs import backend_ordering_matrix, relerr

pytestmark = pytest.mark.gpu


def matern(nx=20, ny=20, seed=3):
    m = spde.grid_mesh_2d(nx, ny, jitter=0.2, seed=seed)
    return m, sp.csc_matrix(spde.matern_precision(m, 0, 0.4))


# ---------------------------------------------------------------------------------------------- seam A
def test_capability_traits_and_algorithm_resolution():
    alg = ls.MI355XCholesky()
    assert ls.supports_selinv(alg) and ls.supports_backward_solve(alg)
    assert not ls.supports_selinv(object()) and not ls.supports_backward_solve(object())
    _, Q = matern(8, 8)
    A, ralg = ls.resolve_linsolve(Q, alg)
    assert isinstance(A, ls.Symmetric) and ralg is alg                 # sparse precision: algorithm honoured
    A2, ralg2 = ls.resolve_linsolve(Q.toarray(), alg)                  # dense storage: dropped for LinearSolve's default
    assert ralg2 is None
    assert ls.configure_algorithm(alg) is alg


def test_gmrf_on_seam_a_matches_dense_identities():
    """test_linearsolve_architecture.jl:61-75 / test_gmrf.jl:36-76 with alg = MI355XCholesky()."""
    _, Q = matern()
    n = Q.shape[0]
    Qd = Q.toarray()
    Sigma = np.linalg.inv(Qd)
    rng = np.random.default_rng(0)
    mu = rng.standard_normal(n)
    d = ls.GMRF(mu, Q, ls.MI355XCholesky())
    assert d.linsolve_cache.isfresh                                    # init does not factorise
    assert np.allclose(ls.var(d), np.diag(Sigma), rtol=1e-8)
    assert not d.linsolve_cache.isfresh                                # ensure_factorization! ran solve! once
    assert np.allclose(ls.std(d), np.sqrt(np.diag(Sigma)), rtol=1e-8)
    assert abs(ls.logdetcov(d) + np.linalg.slogdet(Qd)[1]) < 1e-10 * abs(np.linalg.slogdet(Qd)[1])
    Z = ls.selinv(d.linsolve_cache)
    mask = (Q != 0).toarray()
    assert np.allclose(Z.toarray()[mask], Sigma[mask], rtol=1e-6, atol=1e-12)
    x = rng.standard_normal(n)
    r = x - mu
    assert abs(ls.sqmahal(d, x) - r @ Qd @ r) < 1e-10 * abs(r @ Qd @ r)
    assert np.allclose(ls.gradlogpdf(d, x), -Qd @ r, rtol=1e-12, atol=1e-12)
    assert np.allclose(ls.gradlogpdf(d, mu), 0.0)
    lp = -0.5 * (n * np.log(2 * np.pi) - np.linalg.slogdet(Qd)[1] + r @ Qd @ r)
    assert abs(ls.logpdf(d, x) - lp) < 1e-10 * abs(lp)
    # rand: moments at 5e4 draws (test_gmrf_workspace.jl:85-100 tolerances), batched in one sweep
    S = ls.rand(np.random.default_rng(7), d, 50000)
    assert np.allclose(S.mean(axis=1), mu, atol=5 * np.sqrt(np.diag(Sigma).max() / 50000))
    assert np.allclose(S.var(axis=1), np.diag(Sigma), rtol=0.1)
    one = ls.rand(np.random.default_rng(7), d)
    assert one.shape == (n,)
    # information-vector constructor solves Q mu = h (gmrf.jl:195-223)
    h = rng.standard_normal(n)
    di = ls.GMRF(ls.InformationVector(h), Q, ls.MI355XCholesky())
    assert relerr(di.mean, np.linalg.solve(Qd, h)) < 1e-10
    assert not di.linsolve_cache.isfresh
    with pytest.raises(ValueError):
        ls.GMRF(np.zeros(n + 1), Q, ls.MI355XCholesky())


def test_cache_protocol_mutation_refactor_and_deepcopy():
    """cache.b / cache.A mutation, isfresh, deepcopy(cache) -> gmrfx_clone (gaussian_approximation.jl:61-125)."""
    _, Q = matern(16, 16)
    n = Q.shape[0]
    Qd = Q.toarray()
    rng = np.random.default_rng(1)
    b1, b2 = rng.standard_normal(n), rng.standard_normal(n)
    cache = ls.init(ls.LinearProblem(ls.Symmetric(Q), b1.copy()), ls.MI355XCholesky())
    assert cache.isfresh and cache.cacheval.be is None
    forked_early = copy.deepcopy(cache)                                # deepcopy before the first solve!
    assert relerr(ls.solve(cache).u, np.linalg.solve(Qd, b1)) < 1e-10
    assert not cache.isfresh
    be0 = cache.cacheval.be
    t_factor = be0.stats()["ms_factor"]
    cache.b = b2                                                       # new right-hand side: NO refactorisation
    assert not cache.isfresh
    assert relerr(ls.solve(cache).u, np.linalg.solve(Qd, b2)) < 1e-10
    assert cache.cacheval.be is be0 and be0.stats()["ms_factor"] == t_factor
    fork = copy.deepcopy(cache)                                        # forks the factor (Newton loops)
    assert fork.cacheval.be is not be0 and not fork.isfresh
    Q2 = Q.copy(); Q2.data *= 3.0
    cache.A = ls.Symmetric(Q2)                                         # same pattern, new values: numeric refactorisation
    assert cache.isfresh
    assert relerr(ls.solve(cache).u, np.linalg.solve(3.0 * Qd, b2)) < 1e-10
    assert cache.cacheval.be is be0                                    # symbolic analysis reused
    assert abs(ls.logdet_cov(cache) + np.linalg.slogdet(3.0 * Qd)[1]) < 1e-9
    assert relerr(ls.solve(fork).u, np.linalg.solve(Qd, b2)) < 1e-10   # the fork still holds the OLD factor
    assert abs(ls.logdet_cov(fork) + np.linalg.slogdet(Qd)[1]) < 1e-9
    assert relerr(ls.solve(forked_early).u, np.linalg.solve(Qd, b1)) < 1e-10
    # a different PATTERN through cache.A (storage type carried over, structure not): new symbolic analysis
    Q3 = sp.csc_matrix(Q + sp.diags([0.01 * np.ones(n - 7)] * 2, [7, -7]))
    assert Q3.nnz != Q.nnz
    cache.A = ls.Symmetric(Q3)
    assert relerr(ls.solve(cache).u, np.linalg.solve(Q3.toarray(), b2)) < 1e-10
    assert cache.cacheval.be is not be0
    # seam A throws on an indefinite matrix (CHOLMOD's PosDefException through solve!)
    Qn = Q.copy(); Qn.data *= -1.0
    bad = ls.init(ls.LinearProblem(ls.Symmetric(Qn), b1.copy()), ls.MI355XCholesky())
    with pytest.raises(ls.PosDefException):
        ls.solve(bad)


def _dense_newton(Qd, mu, y, iters, A=None):
    """Dense restatement of the Newton iteration for a Poisson (log link) likelihood: H = -diag(exp(x))."""
    x = mu.copy()
    xs = []
    for _ in range(iters):
        H = -np.diag(np.exp(x))
        g = y - np.exp(x)
        Qn = Qd - H
        x_new = np.linalg.solve(Qn, Qd @ mu + g - H @ x)
        if A is not None:
            step = x_new - x
            At = np.linalg.solve(Qn, A.T)
            step = step - At @ np.linalg.solve(A @ At, A @ step)
            x_new = x + step
        x = x_new
        xs.append(x.copy())
    return xs, Qd + np.diag(np.exp(x))


@pytest.mark.parametrize("constrained", [False, True])
def test_newton_loop_on_the_cache_matches_dense_newton(constrained):
    """_newton_loop (gaussian_approximation.jl:428-499): per iterate `_ga_refactor!` (cache.A = Q_prior - H) and
    `_ga_solve`, optional `_constrain_step` with m column solves, final posterior GMRF on the same cache."""
    _, Q = matern(14, 14, seed=5)
    n = Q.shape[0]
    Qd = Q.toarray()
    rng = np.random.default_rng(2)
    mu = np.zeros(n)
    y = rng.poisson(np.exp(0.3 * rng.standard_normal(n))).astype(float)
    A = np.ones((1, n)) if constrained else None
    prior = ls.GMRF(mu, Q, ls.MI355XCholesky())
    solver = ls._ga_resolve_cache(prior.linsolve_cache, None)          # deepcopy of the prior's cache
    x = mu.copy()
    xs = []
    for _ in range(8):
        H = sp.diags(-np.exp(x)).tocsc()
        Q_new = ls._ga_refactor(solver, Q, H)
        assert solver.isfresh
        x_new = ls._ga_solve(solver, Q @ mu + (y - np.exp(x)) - H @ x)
        if constrained:
            x_new = x + ls._constrain_step(x_new - x, solver, {"A": sp.csr_matrix(A)})
        x = x_new
        xs.append(x.copy())
    xd, Qpost = _dense_newton(Qd, mu, y, 8, A)
    for a, b in zip(xs, xd):
        assert relerr(a, b) < 1e-9
    if constrained:
        assert abs(x.sum()) < 1e-9
    # _build_posterior: refresh the factorisation at the mode, posterior GMRF reuses the solver cache
    ls._ga_refactor(solver, Q, sp.diags(-np.exp(x)).tocsc())
    post = ls.GMRF(x, Q_new.parent, linsolve_cache=solver)
    assert np.allclose(ls.var(post), np.diag(np.linalg.inv(Qpost)), rtol=1e-7)
    assert prior.linsolve_cache.isfresh or relerr(ls.var(prior), np.diag(np.linalg.inv(Qd))) < 1e-8   # the prior's cache is untouched


# ---------------------------------------------------------------------------------------------- ordering forms
def test_ordering_forms_do_not_change_answers():
    """test_backend_ordering.jl:33-60 on its own 145 x 145 matrix: algorithm object, PinDenseColumns, pool."""
    _, Q = backend_ordering_matrix()
    N = Q.shape[0]
    rhs = np.random.default_rng(0).standard_normal(N)
    ws0 = GMRFWorkspace(Q)
    x0, ld0 = ws0.backend.backend_solve(rhs), ws0.backend.compute_logdet()
    ws0.ensure_selinv()
    d0 = ws0.backend.get_selinv_diag()
    for ordering in (mmd, PinDenseColumns(mmd), PinDenseColumns(), PinDenseColumns(np.arange(N - 2, -1, -1))):
        ws = GMRFWorkspace(Q, ordering=ordering)
        assert relerr(ws.backend.backend_solve(rhs), x0) < 1e-10
        assert abs(ws.backend.compute_logdet() - ld0) < 1e-10 * abs(ld0)
        ws.ensure_selinv()
        assert relerr(ws.backend.get_selinv_diag(), d0) < 1e-8
        if isinstance(ordering, PinDenseColumns):
            assert ws.backend.ordering_permutation()[-1] == N - 1
    # "pool shares one resolved ordering" (:55-60): resolve ONCE, hand the vector to every member
    p = gmrfx.ordering_permutation(Q, PinDenseColumns(mmd))
    pool = WorkspacePool(Q, size=2, ordering=p)
    with pool.with_workspace() as ws:
        assert relerr(ws.backend.backend_solve(rhs), x0) < 1e-10
    # "refactorization keeps the custom symbolic" (:62-68)
    ws = GMRFWorkspace(Q, ordering=mmd)
    Q2 = Q.copy(); Q2.data *= 2.0
    ws.update_precision(Q2)
    ws.ensure_numeric()
    assert abs(ws.backend.compute_logdet() - (ld0 + N * np.log(2.0))) < 1e-9 * abs(ld0)


# ---------------------------------------------------------------------------------------------- constraints
def _dense_constrained(Qd, mu, A, e):
    Sigma = np.linalg.inv(Qd)
    SAt = Sigma @ A.T
    W = A @ SAt
    mean_c = mu - SAt @ np.linalg.solve(W, A @ mu - e)
    Sigma_c = Sigma - SAt @ np.linalg.solve(W, SAt.T)
    return Sigma, SAt, W, mean_c, Sigma_c


@pytest.mark.parametrize("m_constraints", [1, 3])
def test_constraint_info_and_constrained_workspace_gmrf(m_constraints):
    """ConstraintInfo (workspace_gmrf.jl:22-56): ONE n x m multi-RHS solve, m x m Cholesky; constrained mean, var
    (:260-273), rand (:275-286), logpdf (:288-305) against the dense kriging formulas (Rue & Held 2005, 2.3.3)."""
    _, Q = matern(18, 15, seed=8)
    n = Q.shape[0]
    Qd = Q.toarray()
    rng = np.random.default_rng(4)
    mu = rng.standard_normal(n)
    A = np.zeros((m_constraints, n))
    A[0, :] = 1.0                                                     # sum-to-zero
    for k in range(1, m_constraints):
        A[k, rng.choice(n, 12, replace=False)] = rng.standard_normal(12)
    e = np.concatenate([[0.0], rng.standard_normal(m_constraints - 1)])
    ws = GMRFWorkspace(Q)
    nsolves0 = ws.backend.stats()["last_nrhs"]
    d = WorkspaceGMRF(mu, Q, ws, sp.csr_matrix(A), e)
    ci = d.constraints
    assert isinstance(ci, ConstraintInfo)
    assert ws.backend.stats()["last_nrhs"] == m_constraints           # the one blocked multi-RHS solve
    Sigma, SAt, W, mean_c, Sigma_c = _dense_constrained(Qd, mu, A, e)
    assert relerr(ci.A_tilde_T, SAt) < 1e-10
    assert relerr(ci.L_c @ ci.L_c.T, W) < 1e-10
    assert relerr(d.mean(), mean_c) < 1e-10
    assert np.allclose(A @ d.mean(), e, atol=1e-10)
    v = d.var()
    assert np.allclose(v, np.diag(Sigma_c), rtol=1e-7, atol=1e-12) and v.min() >= 0.0
    # unconstrained sibling on the SAME workspace: version ping-pong (test_workspace_gmrf.jl:120-160)
    d0 = WorkspaceGMRF(mu, Q, ws)
    assert np.allclose(d0.var(), np.diag(Sigma), rtol=1e-8)
    assert np.allclose(d.var(), np.diag(Sigma_c), rtol=1e-7, atol=1e-12)
    X = d.rand(np.random.default_rng(11), 20000)
    assert np.abs(A @ X - e[:, None]).max() < 1e-9                    # every sample satisfies the constraints
    assert np.allclose(X.mean(axis=1), mean_c, atol=5 * np.sqrt(np.diag(Sigma_c).max() / 20000) + 1e-12)
    big = np.diag(Sigma_c) > 0.05 * np.diag(Sigma_c).max()
    assert np.allclose(X.var(axis=1)[big], np.diag(Sigma_c)[big], rtol=0.1)
    # logpdf of a point on the constraint set: log pi(x) - log pi(Ax = e) - 0.5 log|AA'|
    x = X[:, 0]
    r = x - mu
    lp = -0.5 * (n * np.log(2 * np.pi) - np.linalg.slogdet(Qd)[1] + r @ Qd @ r)
    re = e - A @ mu
    l_ax = -0.5 * (m_constraints * np.log(2 * np.pi) + np.linalg.slogdet(W)[1] + re @ np.linalg.solve(W, re))
    expect = lp - l_ax - 0.5 * np.linalg.slogdet(A @ A.T)[1]
    assert abs(d.logpdf(x) - expect) < 1e-9 * abs(expect)
    with pytest.raises(ValueError):
        ConstraintInfo(ws, mu, np.ones((1, n + 1)), [0.0])
    with pytest.raises(ValueError):
        ConstraintInfo(ws, mu, np.ones((2, n)), [0.0])


@pytest.mark.parametrize("constrained", [False, True])
def test_batched_rand_equals_the_references_column_loop(constrained, monkeypatch):
    """rand(d, k) through the plug-in (julia/GMRFX.jl: Distributions._rand!(rng, d, X::AbstractMatrix) on both seams) is ONE
    multi-column backward sweep; the reference draws the k columns one by one (gmrf.jl:271-281, workspace_gmrf.jl:275-286).
    Same standard-normal draws -> the same samples to 1e-12: the columns of a sweep never interact, but passes of one to sixteen
    columns take the one-wave kernels (bottom tasks: sweep_wave.hip; level fronts: k_bwd_wave) and wider passes the 64-column
    ones -- the same sums in a different order. (Until round 5 the level kernels were shared, and GMRFX_TASK_MODE=wg made the
    two bit-identical; both task modes are still run.) Moments against the dense covariance; k = 70 crosses the 64-column
    pass boundary."""
    mesh, Q = matern(30, 26, seed=12)
    n = Q.shape[0]
    rng = np.random.default_rng(21)
    mu = rng.standard_normal(n)
    A = e = None
    if constrained:
        A = np.zeros((2, n)); A[0, :] = 1.0; A[1, rng.choice(n, 9, replace=False)] = rng.standard_normal(9)
        e = np.array([0.0, 0.3])
    Z = rng.standard_normal((n, 70))
    for mode in (None, "wg"):
        if mode is None:
            monkeypatch.delenv("GMRFX_TASK_MODE", raising=False)
        else:
            monkeypatch.setenv("GMRFX_TASK_MODE", mode)
        ws = GMRFWorkspace(Q, coords=mesh.points)
        d = WorkspaceGMRF(mu, Q, ws, None if A is None else sp.csr_matrix(A), e)
        Xb = d.rand_from(Z)
        Xc = d.rand_from_column_by_column(Z)
        assert ws.backend.stats()["last_nrhs"] == 1                      # the column loop really went one by one
        assert relerr(Xb, Xc) < 1e-12
        if constrained:
            assert np.abs(A @ Xb - e[:, None]).max() < 1e-9
        # seam A twin: GMRF(mean, Q, MI355XCholesky()): backward_solve(cache, Z::Matrix) = one sweep
        g = ls.GMRF(mu, Q, ls.MI355XCholesky())
        Xa = ls.backward_solve(g.linsolve_cache, Z) + mu[:, None]
        Xa1 = np.stack([ls.backward_solve(g.linsolve_cache, Z[:, j]) + mu for j in range(Z.shape[1])], axis=1)
        assert relerr(Xa, Xa1) < 1e-12
    # moments of the batched sampler (unconstrained: cov = Q^-1)
    if not constrained:
        ws = GMRFWorkspace(Q, coords=mesh.points)
        d = WorkspaceGMRF(np.zeros(n), Q, ws)
        X = d.rand(np.random.default_rng(5), 20000)
        Sig = np.diag(np.linalg.inv(Q.toarray()))
        assert np.allclose(X.var(axis=1), Sig, rtol=0.1)
