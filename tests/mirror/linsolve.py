"""TEST INFRASTRUCTURE: the reference's seam A, call for call, in Python -- the LinearSolve cache protocol as the
reference uses it, the `MI355XCholesky` algorithm of julia/GMRFX.jl, the GMRF-side hooks the Pardiso extension
overloads, and the `GMRF` operations that sit on them. Every numeric result comes from libgmrfx.so through
`gmrfx.MI355XBackend`; nothing here computes with Q.

What is mirrored (reference file:line -> name here):
  LinearSolve.LinearProblem / init / solve! / LinearCache fields A, b, u, alg, cacheval, isfresh
      (used at src/gmrf.jl:186-188, 213-220; arithmetic/condition/gaussian_approximation.jl:61-125)  -> LinearProblem, init, solve
  `cache.A = Q` marks the cache fresh (LinearSolve setproperty!), `cache.b = b` does not              -> LinearCache.__setattr__
  deepcopy(cache) forks the solver state (gaussian_approximation.jl:103-109)                           -> copy.deepcopy(cache)
  src/solvers/utils.jl:9-14        ensure_factorization!                                               -> ensure_factorization
  src/solvers/selinv.jl:16-125     supports_selinv, selinv_diag, selinv                                -> same names
  src/solvers/backward_solve.jl    supports_backward_solve, backward_solve                             -> same names
  src/solvers/logdet.jl:12-31      logdet_cov  (NEGATIVE log det Q)                                    -> logdet_cov
  src/utils/linsolve_utils.jl      prepare_for_linsolve, configure_algorithm, algorithm_applicable, resolve_linsolve
  src/gmrf.jl:159-332, 94-106      GMRF(mean, Q, alg), GMRF(InformationVector, Q, alg), logdetcov, var, std, rand, sqmahal,
                                   gradlogpdf, logpdf
  gaussian_approximation.jl:41-125 _constrain_step, _ga_resolve_cache, _ga_refactor!, _ga_solve
"""
from __future__ import annotations

import copy

import numpy as np
import scipy.sparse as sp

import gmrfx
from gmrfx._lib import PosDefException  # noqa: F401  (seam A throws on indefiniteness)


# ---- LinearSolve side ---------------------------------------------------------------------------------------
class MI355XCacheval:
    """Concrete mutable holder returned by init_cacheval: LinearCache's `cacheval` field is typed on it, the
    backend handle is created on the first solve! (julia/GMRFX.jl `MI355XCacheval`)."""

    def __init__(self):
        self.be = None
        self.pattern = None          # (colptr, rowval) the handle was analysed for

    def __deepcopy__(self, memo):    # Base.deepcopy_internal -> gmrfx_clone: the factor survives deepcopy(cache)
        out = MI355XCacheval()
        memo[id(self)] = out
        if self.be is not None:
            out.be = self.be.clone()
            out.pattern = self.pattern
        return out


class MI355XCholesky:
    """LinearSolve algorithm type (`<: LinearSolve.AbstractFactorization`)."""

    def __init__(self, ordering=None, coords=None, device: int = -1):
        self.ordering, self.coords, self.device = ordering, coords, device

    def init_cacheval(self, A, b):
        return MI355XCacheval()


class Symmetric:
    """`Symmetric(A)` wrapper (uplo = :U): what prepare_for_linsolve hands to LinearSolve."""

    def __init__(self, A):
        self.parent = sp.csc_matrix(A)
        self.shape = self.parent.shape

    def __sub__(self, H):
        return Symmetric(sp.csc_matrix(self.parent - H))


class LinearProblem:
    def __init__(self, A, b):
        self.A, self.b = A, b


class LinearSolution:
    def __init__(self, u):
        self.u = u


class LinearCache:
    def __init__(self, A, b, alg):
        object.__setattr__(self, "isfresh", True)
        object.__setattr__(self, "A", A)
        self.b = np.array(b, dtype=np.float64)
        self.u = np.zeros_like(self.b)
        self.alg = alg
        self.cacheval = alg.init_cacheval(A, b)

    def __setattr__(self, name, value):
        if name == "A":                       # LinearSolve: assigning A invalidates the factorisation
            object.__setattr__(self, "isfresh", True)
        object.__setattr__(self, name, value)


def init(prob: LinearProblem, alg) -> LinearCache:
    if alg is None:
        raise NotImplementedError("the mirror only carries the MI355XCholesky algorithm (LinearSolve's defaults stay in Julia)")
    return LinearCache(prob.A, prob.b, alg)


def solve(cache: LinearCache) -> LinearSolution:
    """SciMLBase.solve!(cache, ::MI355XCholesky) of julia/GMRFX.jl."""
    alg = cache.alg
    A = cache.A.parent if isinstance(cache.A, Symmetric) else sp.csc_matrix(cache.A)
    if not A.has_sorted_indices:
        A = A.copy(); A.sort_indices()
    cv = cache.cacheval
    if cache.isfresh:
        pat = (A.indptr, A.indices)
        same = cv.be is not None and cv.pattern[0].shape == pat[0].shape and cv.pattern[1].shape == pat[1].shape \
            and np.array_equal(cv.pattern[0], pat[0]) and np.array_equal(cv.pattern[1], pat[1])
        if not same:                          # first solve, or cache.A got a new pattern: new symbolic analysis
            if cv.be is not None:
                cv.be.close()
            cv.be = gmrfx.MI355XBackend(A, ordering=alg.ordering, coords=alg.coords, device=alg.device,
                                        check_posdef=True, factorize=False)       # seam A throws PosDefException
            cv.pattern = (A.indptr.copy(), A.indices.copy())
        cv.be.refactorize_values(A.data)
        cache.isfresh = False
    cache.u[...] = cv.be.backend_solve(cache.b)
    return LinearSolution(cache.u)


# ---- GMRF-side hooks (what ext/GaussianMarkovRandomFieldsPardiso.jl overloads for PardisoJL) ------------------
def supports_selinv(alg) -> bool:
    return isinstance(alg, MI355XCholesky)


def supports_backward_solve(alg) -> bool:
    return isinstance(alg, MI355XCholesky)


def configure_algorithm(alg):
    return alg


class DenseSymmetric:
    """`Symmetric(::Matrix)`: what the generic prepare_for_linsolve (symmetrize) makes of a dense precision."""

    def __init__(self, A):
        self.parent = np.asarray(A)
        self.shape = self.parent.shape


def prepare_for_linsolve(A, alg):
    if isinstance(A, (Symmetric, DenseSymmetric)):
        return A
    return Symmetric(A) if sp.issparse(A) else DenseSymmetric(A)


def algorithm_applicable(alg, A) -> bool:
    """Val{true} only for sparse storage (a dense or SymTridiagonal precision falls back to LinearSolve's default)."""
    if isinstance(alg, MI355XCholesky):
        return isinstance(A, Symmetric) or sp.issparse(A)
    return True


def resolve_linsolve(precision, alg):
    alg = configure_algorithm(alg)
    A = prepare_for_linsolve(precision, alg)
    if algorithm_applicable(alg, A):
        return A, alg
    return prepare_for_linsolve(precision, None), None


def ensure_factorization(cache: LinearCache) -> None:
    if cache.isfresh:
        solve(cache)


def _get_cacheval(cache) -> gmrfx.MI355XBackend:      # LinearSolve.@get_cacheval(cache, :MI355XCholesky).be
    return cache.cacheval.be


def selinv_diag(cache):
    ensure_factorization(cache)
    return _get_cacheval(cache).get_selinv_diag()


def selinv(cache):
    ensure_factorization(cache)
    return _get_cacheval(cache).get_selinv()          # Symmetric(sparse(Z)): both triangles stored, original ordering


def backward_solve(cache, x):
    ensure_factorization(cache)
    return _get_cacheval(cache).backend_backward_solve(np.array(x, dtype=np.float64))     # collect(): views are copied


def logdet_cov(cache) -> float:
    ensure_factorization(cache)
    return -_get_cacheval(cache).compute_logdet()     # note the sign (solvers/logdet.jl:30)


# ---- GMRF on seam A (src/gmrf.jl) -----------------------------------------------------------------------------
class InformationVector:
    def __init__(self, data):
        self.data = np.asarray(data, dtype=np.float64)


class GMRF:
    def __init__(self, mean_or_information, precision, alg=None, linsolve_cache=None):
        Q = precision.parent if isinstance(precision, Symmetric) else sp.csc_matrix(precision)
        n = Q.shape[0]
        info = isinstance(mean_or_information, InformationVector)
        v = mean_or_information.data if info else np.asarray(mean_or_information, dtype=np.float64)
        if not (v.shape[0] == n == Q.shape[1]):
            raise ValueError("size mismatch")                       # ArgumentError, gmrf.jl:168
        self.precision = Q
        if linsolve_cache is None:
            A, ralg = resolve_linsolve(Q, alg)
            linsolve_cache = init(LinearProblem(A, v.copy()), ralg)
        elif info:
            linsolve_cache.b[...] = v
        self.linsolve_cache = linsolve_cache
        if info:
            self.information_vector = v
            self.mean = solve(linsolve_cache).u.copy()              # gmrf.jl:220
        else:
            self.information_vector = None
            self.mean = v

    def __len__(self):
        return self.precision.shape[0]


def logdetcov(d: GMRF) -> float:
    return logdet_cov(d.linsolve_cache)


def var(d: GMRF):
    assert supports_selinv(d.linsolve_cache.alg)
    return np.array(selinv_diag(d.linsolve_cache))


def std(d: GMRF):
    return np.sqrt(var(d))


def rand(rng: np.random.Generator, d: GMRF, k: int | None = None):
    """`rand(rng, d)` / `rand(rng, d, k)`: z ~ N(0, I), x = P' L^-T z + mean (gmrf.jl:271-281). The reference solves the
    k columns one by one; the backend takes them as one batch (same values: columns do not interact)."""
    n = len(d)
    z = rng.standard_normal(n if k is None else (n, k))
    x = backward_solve(d.linsolve_cache, z)
    return x + (d.mean if k is None else d.mean[:, None])


def sqmahal(d: GMRF, x):
    r = np.asarray(x) - d.mean
    return float(r @ (d.precision @ r))


def gradlogpdf(d: GMRF, x):
    return -(d.precision @ (np.asarray(x) - d.mean))


def logpdf(d: GMRF, x) -> float:
    n = len(d)
    return -0.5 * (n * np.log(2.0 * np.pi) + logdetcov(d) + sqmahal(d, x))


# ---- Newton loop pieces (arithmetic/condition/gaussian_approximation.jl:41-125) --------------------------------
def _ga_resolve_cache(cache: LinearCache, Q_posterior):
    return copy.deepcopy(cache)                       # sparse prior, sparse posterior: the storage type carries over


def _ga_refactor(cache: LinearCache, Q_prior, H):
    Q_new = prepare_for_linsolve(sp.csc_matrix(Q_prior - H), cache.alg)
    cache.A = Q_new                                   # _update_linsolve_cache!: marks the cache fresh
    return Q_new


def _ga_solve(cache: LinearCache, b):
    cache.b = np.array(b, dtype=np.float64)
    return solve(cache).u.copy()


def _constrain_step(step, cache: LinearCache, constraints):
    """KKT projection with m column solves on the current factorisation (gaussian_approximation.jl:41-58)."""
    if constraints is None:
        return step
    A = np.asarray(constraints["A"].todense() if sp.issparse(constraints["A"]) else constraints["A"], dtype=np.float64)
    m, n = A.shape
    A_tilde_T = np.empty((n, m))
    saved_b = cache.b.copy()
    for i in range(m):
        cache.b[...] = A[i, :]
        A_tilde_T[:, i] = solve(cache).u
    cache.b[...] = saved_b
    L_c = np.linalg.cholesky(A @ A_tilde_T)
    y = np.linalg.solve(L_c.T, np.linalg.solve(L_c, A @ step))
    return step - A_tilde_T @ y
