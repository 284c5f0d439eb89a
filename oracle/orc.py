"""ctypes binding of the CPU oracle (oracle/gmrf_oracle.c). TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_factorize.restype = C.c_void_p
        L.orc_factorize.argtypes = [C.c_int64, i64p, i64p, f64p, C.c_int, C.c_void_p]
        L.orc_free.argtypes = [C.c_void_p]
        for name in ("orc_n", "orc_nnz_L", "orc_fail_col", "orc_selinv_nnz"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_get_L.argtypes = [C.c_void_p, i64p, i64p, f64p]
        L.orc_get_parent.argtypes = [C.c_void_p, i64p]
        L.orc_solve.argtypes = [C.c_void_p, f64p, C.c_int64, C.c_int64, f64p, C.c_int64]
        L.orc_backward_solve.argtypes = [C.c_void_p, f64p, C.c_int64, C.c_int64, f64p, C.c_int64]
        L.orc_logdet.restype = C.c_double
        L.orc_logdet.argtypes = [C.c_void_p]
        L.orc_selinv_diag.argtypes = [C.c_void_p, f64p]
        L.orc_selinv_csc.argtypes = [C.c_void_p, i64p, i64p, f64p]
        L.orc_sqmahal.restype = C.c_double
        L.orc_sqmahal.argtypes = [C.c_int64, i64p, i64p, f64p, C.c_int, f64p, C.c_void_p]
        _LIB = L
    return _LIB


def sqmahal(Q, x, mean=None, uplo: str = "U") -> float:
    """(x - mean)' Symmetric(Q, uplo) (x - mean): gmrf.jl:94-97 / workspace_gmrf.jl:288-292."""
    Q = sp.csc_matrix(Q)
    Ap = np.ascontiguousarray(Q.indptr, dtype=np.int64)
    Ai = np.ascontiguousarray(Q.indices, dtype=np.int64)
    Ax = np.ascontiguousarray(Q.data, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    mu = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
    return lib().orc_sqmahal(Q.shape[0], Ap, Ai, Ax, ord(uplo), x, None if mu is None else mu.ctypes.data_as(C.c_void_p))


def logpdf(F: "OracleFactor", Q, z, mean=None, uplo: str = "U") -> float:
    """logpdf(d::WorkspaceGMRF, z), unconstrained: workspace_gmrf.jl:288-292 (logdetcov = -logdet Q)."""
    n = Q.shape[0]
    return -0.5 * sqmahal(Q, z, mean, uplo) + 0.5 * F.logdet() - 0.5 * n * np.log(2.0 * np.pi)


class OracleFactor:
    """cholesky(Symmetric(Q); perm) + the operations the reference calls on it."""

    def __init__(self, Q: sp.spmatrix, perm=None, uplo: str = "U"):
        Q = sp.csc_matrix(Q)
        Q.sort_indices()
        self.n = Q.shape[0]
        Ap = np.ascontiguousarray(Q.indptr, dtype=np.int64)
        Ai = np.ascontiguousarray(Q.indices, dtype=np.int64)
        Ax = np.ascontiguousarray(Q.data, dtype=np.float64)
        pp = None
        if perm is not None:
            self._perm = np.ascontiguousarray(perm, dtype=np.int64)
            pp = self._perm.ctypes.data_as(C.c_void_p)
        self._h = lib().orc_factorize(self.n, Ap, Ai, Ax, ord(uplo), pp)
        if not self._h:
            raise ValueError("invalid permutation")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free(self._h)
            self._h = None

    @property
    def nnz_L(self):
        return lib().orc_nnz_L(self._h)

    @property
    def fail_col(self):
        return lib().orc_fail_col(self._h)

    def L(self) -> sp.csc_matrix:
        nz = self.nnz_L
        Lp = np.empty(self.n + 1, np.int64); Li = np.empty(nz, np.int64); Lx = np.empty(nz)
        lib().orc_get_L(self._h, Lp, Li, Lx)
        return sp.csc_matrix((Lx, Li, Lp), shape=(self.n, self.n))

    def parent(self):
        p = np.empty(self.n, np.int64)
        lib().orc_get_parent(self._h, p)
        return p

    def solve(self, B):
        B = np.asarray(B, dtype=np.float64)
        vec = B.ndim == 1
        Bf = np.asfortranarray(B.reshape(self.n, -1))
        X = np.empty_like(Bf, order="F")
        lib().orc_solve(self._h, Bf.T.reshape(-1), self.n, Bf.shape[1], X.T.reshape(-1), self.n)
        return X[:, 0].copy() if vec else X

    def backward_solve(self, Z):
        Z = np.asarray(Z, dtype=np.float64)
        vec = Z.ndim == 1
        Zf = np.asfortranarray(Z.reshape(self.n, -1))
        X = np.empty_like(Zf, order="F")
        lib().orc_backward_solve(self._h, Zf.T.reshape(-1), self.n, Zf.shape[1], X.T.reshape(-1), self.n)
        return X[:, 0].copy() if vec else X

    def logdet(self) -> float:
        return lib().orc_logdet(self._h)

    def selinv_diag(self):
        out = np.empty(self.n)
        lib().orc_selinv_diag(self._h, out)
        return out

    def selinv(self) -> sp.csc_matrix:
        nz = lib().orc_selinv_nnz(self._h)
        Zp = np.empty(self.n + 1, np.int64); Zi = np.empty(nz, np.int64); Zv = np.empty(nz)
        lib().orc_selinv_csc(self._h, Zp, Zi, Zv)
        return sp.csc_matrix((Zv, Zi, Zp), shape=(self.n, self.n))


def row_diag_ASigmaAt(F: "OracleFactor", A) -> np.ndarray:
    """v[i] = sum_{p,q} A[i,p] A[i,q] Sigma[p,q] with Sigma the selected inverse (0 outside the factor pattern):
    the loop of _row_diag_AΣAt, /root/reference/src/linear_predictor_marginals.jl:145-159, restated."""
    A = sp.csr_matrix(A)
    Sig = F.selinv().tocsr()
    Sig.sort_indices()
    out = np.zeros(A.shape[0])
    for i in range(A.shape[0]):
        lo, hi = A.indptr[i], A.indptr[i + 1]
        s = 0.0
        for p in range(lo, hi):
            for q in range(lo, hi):
                s += A.data[p] * A.data[q] * Sig[A.indices[p], A.indices[q]]
        out[i] = s
    return out


def selinv_dot(F: "OracleFactor", B) -> float:
    """dot(selinv(Q), B) = tr(Q^-1 B): /root/reference/src/workspace/backend.jl:258-267 (Sigma = 0 outside
    the factor pattern)."""
    B = sp.csc_matrix(B)
    Sig = F.selinv().tocsc()
    return float(Sig.multiply(B).sum())


# ---- KL (Vecchia) sparse approximate Cholesky: /root/reference/src/kl_cholesky/kl_cholesky.jl ----------------------

def kl_cholesky_inplace(Theta, L, reg: float = 1e-6) -> sp.csc_matrix:
    """sparse_approximate_cholesky!(Theta, L), kl_cholesky.jl:32-55, restated loop by loop: per column k the row
    indices in reverse order, M = Theta[S, S] + reg I = U'U (cholesky!), U x = e_last (ldiv!), L[S, k] = x."""
    import scipy.linalg as sl
    Theta = np.asarray(Theta, dtype=np.float64)
    L = sp.csc_matrix(L, dtype=np.float64).copy()
    L.sort_indices()
    for k in range(L.shape[1]):
        idx = np.arange(L.indptr[k], L.indptr[k + 1])[::-1]
        S = L.indices[idx]
        M = Theta[np.ix_(S, S)] + reg * np.eye(len(S))
        U = sl.cholesky(M, lower=False)
        x = np.zeros(len(S)); x[-1] = 1.0
        x = sl.solve_triangular(U, x, lower=False)
        L.data[idx] = x
    return L


def kl_cholesky_supernodal(Theta, column_indices, row_indices, reg: float = 1e-8) -> sp.csc_matrix:
    """sparse_approximate_cholesky(Theta, sc), kl_cholesky.jl:74-113 (+ the pattern of :57-72), restated: per
    supernode one Cholesky of Theta[R, R] + reg I (R as given: descending), per member column k the right-hand
    side e_{N_k}, N_k = nnz(L[:, k]), and L.nzval[column k] = x[N_k:-1:1]."""
    import scipy.linalg as sl
    Theta = np.asarray(Theta, dtype=np.float64)
    n = Theta.shape[0]
    Is, Js = [], []
    for cols, rows in zip(column_indices, row_indices):
        for j in cols:
            for i in rows:
                if j <= i:
                    Is.append(i); Js.append(j)
    L = sp.csc_matrix((np.ones(len(Is)), (Is, Js)), shape=(n, n))
    L.sum_duplicates(); L.sort_indices()
    for cols, rows in zip(column_indices, row_indices):
        R = np.asarray(rows)
        M = Theta[np.ix_(R, R)] + reg * np.eye(len(R))
        U = sl.cholesky(M, lower=False)
        X = np.zeros((len(R), len(cols)))
        for q, k in enumerate(cols):
            X[L.indptr[k + 1] - L.indptr[k] - 1, q] = 1.0
        X = sl.solve_triangular(U, X, lower=False)
        for q, k in enumerate(cols):
            nk = L.indptr[k + 1] - L.indptr[k]
            L.data[L.indptr[k]:L.indptr[k + 1]] = X[:nk, q][::-1]
    return L
