"""MI355XBackend -- mirror of the reference's `WorkspaceBackend` protocol
(src/workspace/backend.jl:8-30) for the libgmrfx.so backend. Method names, argument meaning,
laziness/caching and error behaviour follow `CHOLMODBackend` (backend.jl:51-284) so that the
parity tests read like test/workspace/test_gmrf_workspace.jl. All arithmetic is done by the HIP
kernels behind the C ABI; this file only marshals arrays."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import GmrfxOpts, GmrfxStats, check, lib, ptr


@dataclass
class SymbolicInfo:
    super_first: np.ndarray
    super_parent: np.ndarray
    row_ptr: np.ndarray
    rows: np.ndarray
    rel: np.ndarray
    panel_ptr: np.ndarray
    panel_ld: np.ndarray
    level: np.ndarray
    q_src: np.ndarray
    q_dst: np.ndarray
    cb_arena: int


def _as_csc(Q) -> sp.csc_matrix:
    if not sp.isspmatrix_csc(Q):
        Q = sp.csc_matrix(Q)
    if Q.shape[0] != Q.shape[1]:
        raise ValueError("Q must be square")      # ArgumentError in gmrf_workspace.jl:68
    if not Q.has_sorted_indices:
        Q = Q.copy()
        Q.sort_indices()
    return Q


class MI355XBackend:
    """`MI355XBackend(Symmetric(Q); ordering=nothing, coords=nothing)`.

    ordering: None (own nested dissection), "natural", an explicit permutation vector (0-based here; the Julia
    shim passes 1-based with index_base=1), a callable pattern -> permutation (where Julia passes a CliqueTrees
    algorithm object) or `PinDenseColumns(inner, frac)` -- resolved by `gmrfx.ordering.ordering_permutation`, the
    host logic of `CHOLMODBackend(Q; ordering)` / `ordering_permutation`, backend.jl:86-153. Unknown forms raise.
    """

    def __init__(self, Q, ordering=None, coords=None, device: int = -1, symbolic_only: bool = False,
                 check_posdef: bool = False, uplo: str = "U", nd_leaf: int = 0, relax_cols: int = 0,
                 relax_zeros: float = 0.0, factorize: bool = True, shard_rank: int = 0, shard_world: int = 1, shard_min_top: int = 0):
        Q = _as_csc(Q)
        self.n = Q.shape[0]
        self._colptr = np.ascontiguousarray(Q.indptr, dtype=np.int64)
        self._rowval = np.ascontiguousarray(Q.indices, dtype=np.int64)
        self._nnz = int(self._colptr[-1])
        opts = GmrfxOpts()
        opts.struct_size = C.sizeof(GmrfxOpts)
        opts.uplo = 0 if uplo.upper().startswith("U") else 1
        opts.device = device
        self.device = device            # HIP device ordinal the handle was asked for (-1: the current device)
        opts.symbolic_only = int(symbolic_only)
        opts.check_posdef = int(check_posdef)
        opts.nd_leaf = nd_leaf
        opts.relax_cols = relax_cols
        opts.relax_zeros = relax_zeros
        opts.shard_rank = shard_rank
        opts.shard_world = shard_world
        opts.shard_min_top = shard_min_top      # > 0 with shard_world == 1: a sharded handle of ONE rank (include/gmrfx.h)
        self.shard_rank, self.shard_world = shard_rank, shard_world
        from .ordering import ordering_permutation as _resolve
        perm = _resolve(Q, ordering, coords)
        if isinstance(perm, str):           # "natural"
            opts.ordering = 1
            perm = None
        self._coords = None
        if coords is not None:
            self._coords = np.ascontiguousarray(coords, dtype=np.float64)
            if self._coords.ndim != 2 or self._coords.shape[0] != self.n:
                raise ValueError("coords must be n x dim")
            opts.coord_dim = self._coords.shape[1]
            opts.coords = self._coords.ctypes.data
        h = C.c_void_p()
        check(lib().gmrfx_create(self.n, ptr(self._colptr), ptr(self._rowval), 0, ptr(perm), C.byref(opts), C.byref(h)))
        self._h = h
        self.symbolic_only = symbolic_only
        self._selinv_cache = None
        self._selinv_diag_cache = None
        self.last_info = 0
        if factorize and not symbolic_only and shard_world <= 1 and shard_min_top <= 0:
            self.refactorize_values(Q.data)

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().gmrfx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone(self) -> "MI355XBackend":
        """deepcopy(cache) semantics (arithmetic/condition/gaussian_approximation.jl:103-109)."""
        other = object.__new__(MI355XBackend)
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_h", "_rd_plans")})   # plans live in the handle
        h = C.c_void_p()
        check(lib().gmrfx_clone(self._h, C.byref(h)))
        other._h = h
        other._selinv_cache = None
        other._selinv_diag_cache = None
        return other

    # -- WorkspaceBackend protocol ----------------------------------------------------------
    def refactorize(self, Q) -> None:
        """refactorize!(b, Q::Symmetric): same pattern guaranteed by the caller (backend.jl:178)."""
        Q = _as_csc(Q)
        if Q.nnz != self._nnz:
            raise ValueError(f"buffer holds {self._nnz} values but Q has {Q.nnz} nonzeros; "
                             "the sparsity pattern must be invariant across refactorizations")
        self.refactorize_values(Q.data)

    def refactorize_values(self, nzval) -> None:
        nz = np.ascontiguousarray(nzval, dtype=np.float64)
        if nz.shape != (self._nnz,):
            raise ValueError("nzval length does not match the pattern")
        info = C.c_int64(0)
        check(lib().gmrfx_refactorize(self._h, ptr(nz), C.byref(info)), self._h)
        self.last_info = info.value
        self._selinv_cache = None
        self._selinv_diag_cache = None

    def backend_solve(self, rhs):
        """`F \\ b` / `F \\ B` (backend.jl:191-209). Returns a fresh array; input untouched."""
        B = np.asarray(rhs, dtype=np.float64)
        if B.shape[0] != self.n:
            raise ValueError("dimension mismatch")
        vec = B.ndim == 1
        Bf = np.asfortranarray(B.reshape(self.n, -1))
        X = np.empty_like(Bf, order="F")
        check(lib().gmrfx_solve(self._h, Bf.ctypes.data, self.n, Bf.shape[1], X.ctypes.data, self.n), self._h)
        return X[:, 0].copy() if vec else X

    def compute_logdet(self) -> float:
        out = C.c_double(0.0)
        check(lib().gmrfx_logdet(self._h, C.byref(out)), self._h)
        return out.value

    def sqmahal(self, x, mean=None, nzval=None):
        """(x - mean)' Q (x - mean) on the device: `dot(r, d.precision * r)` of logpdf (workspace_gmrf.jl:288-292)
        and sqmahal (gmrf.jl:94-97). x: n vector or n x k matrix (one value per column). nzval: Q's values in the
        pattern's order; None = the values of the last refactorisation."""
        X = np.asarray(x, dtype=np.float64)
        if X.shape[0] != self.n:
            raise ValueError("dimension mismatch")
        vec = X.ndim == 1
        Xf = np.asfortranarray(X.reshape(self.n, -1))
        mu = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
        if mu is not None and mu.shape != (self.n,):
            raise ValueError("dimension mismatch")
        nz = None if nzval is None else np.ascontiguousarray(nzval, dtype=np.float64)
        if nz is not None and nz.shape != (self._nnz,):
            raise ValueError("nzval length does not match the pattern")
        out = np.empty(Xf.shape[1])
        check(lib().gmrfx_quadform(self._h, ptr(nz), ptr(Xf), self.n, Xf.shape[1], ptr(mu), ptr(out)), self._h)
        return float(out[0]) if vec else out

    def logpdf(self, z, mean=None) -> float:
        """logpdf(d::WorkspaceGMRF, z) without constraints (workspace_gmrf.jl:288-292):
        -0.5 r'Qr + 0.5 logdet(Q) - 0.5 n log(2 pi), on the values of the last refactorisation."""
        return -0.5 * self.sqmahal(z, mean) + 0.5 * self.compute_logdet() - 0.5 * self.n * np.log(2.0 * np.pi)

    def compute_selinv(self) -> None:
        """Lazy like the CHOLMOD backend (backend.jl:215-221): the getters trigger the work."""
        return None

    def get_selinv(self) -> sp.csc_matrix:
        if self._selinv_cache is None:
            nnz = C.c_int64(0)
            check(lib().gmrfx_selinv_nnz(self._h, C.byref(nnz)), self._h)
            colptr = np.empty(self.n + 1, np.int64)
            rowval = np.empty(nnz.value, np.int64)
            nzval = np.empty(nnz.value, np.float64)
            check(lib().gmrfx_selinv_csc(self._h, 0, ptr(colptr), ptr(rowval), ptr(nzval)), self._h)
            self._selinv_cache = sp.csc_matrix((nzval, rowval, colptr), shape=(self.n, self.n))
        return self._selinv_cache

    def get_selinv_diag(self) -> np.ndarray:
        if self._selinv_diag_cache is None:
            out = np.empty(self.n)
            check(lib().gmrfx_selinv_diag(self._h, ptr(out)), self._h)
            self._selinv_diag_cache = out
        return self._selinv_diag_cache

    def backend_backward_solve(self, x):
        """`F.UP \\ x` = P' L^-T x (backend.jl:281-284); accepts views, vectors or n x k."""
        Z = np.asarray(x, dtype=np.float64)
        if Z.shape[0] != self.n:
            raise ValueError("dimension mismatch")
        vec = Z.ndim == 1
        Zf = np.asfortranarray(Z.reshape(self.n, -1))
        X = np.empty_like(Zf, order="F")
        check(lib().gmrfx_backward_solve(self._h, Zf.ctypes.data, self.n, Zf.shape[1], X.ctypes.data, self.n), self._h)
        return X[:, 0].copy() if vec else X

    def selinv_extract_at(self, B) -> sp.csc_matrix:
        """Sigma on B's pattern, 0 outside the factor pattern (backend.jl:275-279)."""
        B = _as_csc(B)
        if B.shape != (self.n, self.n):
            raise ValueError("dimension mismatch")
        colptr = np.ascontiguousarray(B.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(B.indices, dtype=np.int64)
        out = np.empty(len(rowval))
        check(lib().gmrfx_selinv_extract(self._h, self.n, ptr(colptr), ptr(rowval), 0, ptr(out)), self._h)
        return sp.csc_matrix((out, rowval.copy(), colptr.copy()), shape=B.shape)

    def selinv_dot(self, B) -> float:
        """tr(Q^-1 B) for pattern(B) within the factor pattern (backend.jl:265). The values of
        Sigma are gathered on the device; the final dot stays on the host so that, as in the
        reference, B may carry dual numbers."""
        B = _as_csc(B)
        return float(np.dot(self.selinv_extract_at(B).data, B.data))

    def selinv_dot_device(self, B) -> float:
        """tr(Q^-1 B) for a Float64 B, contracted on the device (gmrfx_selinv_dot)."""
        B = _as_csc(B)
        if B.shape != (self.n, self.n):
            raise ValueError("dimension mismatch")
        colptr = np.ascontiguousarray(B.indptr, dtype=np.int64)
        rowval = np.ascontiguousarray(B.indices, dtype=np.int64)
        vals = np.ascontiguousarray(B.data, dtype=np.float64)
        out = C.c_double(0.0)
        check(lib().gmrfx_selinv_dot(self._h, self.n, ptr(colptr), ptr(rowval), ptr(vals), 0, C.byref(out)), self._h)
        return out.value

    def row_diag_ASigmaAt(self, A) -> np.ndarray:
        """diag(A Sigma A') for a sparse design matrix A (m x n): the predictor marginal variances of
        linear_predictor_marginals.jl:125-165, contracted on the device from the selected-inverse panels
        (Sigma = 0 outside the factor pattern, as there). The pair plan of A's pattern is kept on the device and
        reused while the pattern stays the same (the hyper-parameter loop changes Q, not A)."""
        A = sp.csr_matrix(A)
        if A.shape[1] != self.n:
            raise ValueError("dimension mismatch")
        A.sum_duplicates()
        rowptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        colind = np.ascontiguousarray(A.indices, dtype=np.int64)
        vals = np.ascontiguousarray(A.data, dtype=np.float64)
        import zlib
        key = (A.shape[0], len(colind), zlib.crc32(rowptr.tobytes()), zlib.crc32(colind.tobytes()))
        plans = self.__dict__.setdefault("_rd_plans", {})
        if key not in plans:
            if len(plans) >= 4:                                   # a handful of design matrices at most
                old_key = next(iter(plans))
                check(lib().gmrfx_selinv_row_diag_free(self._h, plans.pop(old_key)), self._h)
            pid = C.c_int64(-1)
            check(lib().gmrfx_selinv_row_diag_plan(self._h, A.shape[0], ptr(rowptr), ptr(colind), 0, C.byref(pid)), self._h)
            plans[key] = pid.value
        out = np.empty(A.shape[0])
        check(lib().gmrfx_selinv_row_diag_apply(self._h, plans[key], ptr(vals), ptr(out)), self._h)
        return out

    def row_diag_ASigmaAt_once(self, A) -> np.ndarray:
        """One-shot form (gmrfx_selinv_row_diag): plans, contracts and forgets."""
        A = sp.csr_matrix(A)
        if A.shape[1] != self.n:
            raise ValueError("dimension mismatch")
        A.sum_duplicates()
        rowptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        colind = np.ascontiguousarray(A.indices, dtype=np.int64)
        vals = np.ascontiguousarray(A.data, dtype=np.float64)
        out = np.empty(A.shape[0])
        check(lib().gmrfx_selinv_row_diag(self._h, A.shape[0], ptr(rowptr), ptr(colind), ptr(vals), 0, ptr(out)), self._h)
        return out

    # -- extras --------------------------------------------------------------------------------
    def ordering_permutation(self) -> np.ndarray:
        """Elimination order actually used (0-based): pass it to CHOLMOD to factor the same PQP'."""
        p = np.empty(self.n, np.int64)
        check(lib().gmrfx_get_perm(self._h, 0, ptr(p)))
        return p

    def stats(self) -> dict:
        st = GmrfxStats()
        check(lib().gmrfx_get_stats(self._h, C.byref(st), C.sizeof(GmrfxStats)))
        return st.asdict()

    def symbolic(self) -> SymbolicInfo:
        sizes = np.zeros(8, np.int64)
        check(lib().gmrfx_symbolic_sizes(self._h, ptr(sizes)))
        ns, sr, _, _, cb, nq = (int(x) for x in sizes[:6])
        a = dict(super_first=np.empty(ns + 1, np.int64), super_parent=np.empty(ns, np.int64),
                 row_ptr=np.empty(ns + 1, np.int64), rows=np.empty(sr, np.int64), rel=np.empty(sr, np.int64),
                 panel_ptr=np.empty(ns + 1, np.int64), panel_ld=np.empty(ns, np.int64), level=np.empty(ns, np.int64),
                 q_src=np.empty(nq, np.int64), q_dst=np.empty(nq, np.int64))
        check(lib().gmrfx_symbolic_get(self._h, *[ptr(a[k]) for k in (
            "super_first", "super_parent", "row_ptr", "rows", "rel", "panel_ptr", "panel_ld", "level", "q_src", "q_dst")]))
        return SymbolicInfo(cb_arena=cb, **a)

    def sweep_tasks(self):
        """(rows_cap, first, last, lrow): the bottom subtrees whose sweeps run on an LDS-resident local vector."""
        nt, cap = C.c_int64(0), C.c_int64(0)
        check(lib().gmrfx_symbolic_sweep_tasks(self._h, C.byref(nt), C.byref(cap), None, None, None))
        sizes = np.zeros(8, np.int64)
        check(lib().gmrfx_symbolic_sizes(self._h, ptr(sizes)))
        first, last = np.empty(nt.value, np.int64), np.empty(nt.value, np.int64)
        lrow = np.empty(int(sizes[1]), np.int64)
        check(lib().gmrfx_symbolic_sweep_tasks(self._h, C.byref(nt), C.byref(cap), ptr(first), ptr(last), ptr(lrow)))
        return int(cap.value), first, last, lrow

    def sweep_chunks(self) -> dict:
        """The tasks' chunks (include/gmrfx.h: gmrfx_symbolic_sweep_chunks): task_ptr (ntasks + 1 x 2: first forward / backward
        record), slot (ntasks x 8), fwd / bwd (records x 8: pa, ld, o, cc, nt, lr, nbar, id), rows (padded target-row lists)."""
        nc, nr = np.zeros(2, np.int64), C.c_int64(0)
        check(lib().gmrfx_symbolic_sweep_chunks(self._h, ptr(nc), C.byref(nr), None, None, None, None, None))
        nt = C.c_int64(0)
        check(lib().gmrfx_symbolic_sweep_tasks(self._h, C.byref(nt), None, None, None, None))
        out = dict(task_ptr=np.empty((nt.value + 1, 2), np.int64), slot=np.empty((nt.value, 8), np.int64),
                   fwd=np.empty((int(nc[0]), 8), np.int64), bwd=np.empty((int(nc[1]), 8), np.int64), rows=np.empty(nr.value, np.int64))
        check(lib().gmrfx_symbolic_sweep_chunks(self._h, ptr(nc), C.byref(nr), *[ptr(out[k]) for k in ("task_ptr", "slot", "fwd", "bwd", "rows")]))
        return out

    def factor_values(self) -> np.ndarray:
        sizes = np.zeros(8, np.int64)
        check(lib().gmrfx_symbolic_sizes(self._h, ptr(sizes)))
        out = np.empty(int(sizes[2]))
        check(lib().gmrfx_get_factor_values(self._h, ptr(out)), self._h)
        return out

    def factor_csc(self) -> sp.csc_matrix:
        """The numeric factor L (elimination order) as a sparse matrix, from the panels."""
        sy = self.symbolic()
        vals = self.factor_values()
        I, J, V = [], [], []
        for s in range(len(sy.super_parent)):
            f, l1 = sy.super_first[s], sy.super_first[s + 1]
            rows = sy.rows[sy.row_ptr[s]:sy.row_ptr[s + 1]]
            ld = sy.panel_ld[s]
            for j in range(l1 - f):
                col = vals[sy.panel_ptr[s] + j * ld: sy.panel_ptr[s] + j * ld + len(rows)]
                I.append(rows[j:]); J.append(np.full(len(rows) - j, f + j)); V.append(col[j:])
        return sp.csc_matrix((np.concatenate(V), (np.concatenate(I), np.concatenate(J))), shape=(self.n, self.n))

    # -- device-resident entry points (HBM in, HBM out) for benchmarks / chained GPU use -----------
    def refactorize_dev(self, d_nzval_ptr: int) -> int:
        info = C.c_int64(0)
        check(lib().gmrfx_refactorize_dev(self._h, d_nzval_ptr, C.byref(info)), self._h)
        self._selinv_cache = None
        self._selinv_diag_cache = None
        self.last_info = info.value
        return info.value

    def refactorize_solve_dev(self, d_nzval_ptr: int, d_B: int, ldb: int, nrhs: int, d_X: int, ldx: int) -> int:
        """`workspace_solve` on a workspace with new values (gmrf_workspace.jl:170-178, 207-215): numeric factorisation and
        solve as one pipelined call; same bits as refactorize_dev + solve_dev."""
        info = C.c_int64(0)
        check(lib().gmrfx_refactorize_solve_dev(self._h, d_nzval_ptr, d_B, ldb, nrhs, d_X, ldx, C.byref(info)), self._h)
        self._selinv_cache = None
        self._selinv_diag_cache = None
        self.last_info = info.value
        return info.value

    def refactorize_logpdf_dev(self, d_nzval_ptr: int, d_X: int, ldx: int, nvec: int, d_mu: int = 0):
        """One logpdf evaluation of the hyper-parameter loop in one call (gmrfx_refactorize_logpdf_dev): new values -> numeric
        factorisation; returns (quadratic forms (x_k - mu)' Q (x_k - mu), log det Q). logpdf = -q / 2 + logdet / 2 - n log(2 pi) / 2
        (workspace_gmrf.jl:288-292)."""
        quad = np.empty(max(nvec, 0))
        ld = C.c_double(0.0)
        info = C.c_int64(0)
        check(lib().gmrfx_refactorize_logpdf_dev(self._h, d_nzval_ptr, d_X or None, ldx, nvec, d_mu or None, ptr(quad), C.byref(ld), C.byref(info)), self._h)
        self._selinv_cache = None
        self._selinv_diag_cache = None
        self.last_info = info.value
        return quad, ld.value

    def refactorize_solve(self, nzval, rhs):
        """Host-array form of refactorize_solve_dev: new values of Q (pattern order) and right-hand sides -> X (fresh array)."""
        nz = np.ascontiguousarray(nzval, dtype=np.float64)
        if nz.shape != (self._nnz,):
            raise ValueError("nzval length does not match the pattern")
        B = np.asarray(rhs, dtype=np.float64)
        if B.shape[0] != self.n:
            raise ValueError("dimension mismatch")
        vec = B.ndim == 1
        Bf = np.asfortranarray(B.reshape(self.n, -1))
        X = np.empty_like(Bf, order="F")
        info = C.c_int64(0)
        check(lib().gmrfx_refactorize_solve(self._h, ptr(nz), ptr(Bf), self.n, Bf.shape[1], ptr(X), self.n, C.byref(info)), self._h)
        self._selinv_cache = None
        self._selinv_diag_cache = None
        self.last_info = info.value
        return X[:, 0].copy() if vec else X

    def dense_apply_dev(self, d_D: int, n1: int, d_T: int, n2: int, d_R: int) -> None:
        """R = D T on the device (gmrfx_dense_apply_dev): row-major D (n1 x n1), T and R (n1 x n2); the dense-operator leg of the
        Kronecker path (separable.jl:122-172). T needs 8 readable bytes behind it when n2 is odd."""
        check(lib().gmrfx_dense_apply_dev(self._h, n1, n2, d_D, d_T, d_R), self._h)

    def transpose_dev(self, d_src: int, rows: int, cols: int, d_dst: int) -> None:
        """dst (cols x rows) = src' for a row-major rows x cols device array (gmrfx_transpose_dev)."""
        check(lib().gmrfx_transpose_dev(self._h, rows, cols, d_src, d_dst), self._h)

    def refactorize_solve_ptr(self, nzval_ptr: int, B_ptr: int, ldb: int, nrhs: int, X_ptr: int, ldx: int) -> int:
        """gmrfx_refactorize_solve on raw HOST pointers (column-major B / X the caller keeps alive; page-locked memory is handed to
        the DMA engine directly, pageable memory is staged by the library): the call a Julia `workspace_solve(ws, B::Matrix)` makes."""
        info = C.c_int64(0)
        check(lib().gmrfx_refactorize_solve(self._h, nzval_ptr, B_ptr, ldb, nrhs, X_ptr, ldx, C.byref(info)), self._h)
        self._selinv_cache = None
        self._selinv_diag_cache = None
        self.last_info = info.value
        return info.value

    # -- Newton loop on the device (src/workspace/gaussian_approximation.jl:63-129) --------
    def set_prior(self, prior_nzval, hess_map) -> None:
        """prior_nzval: values of the prior precision in the pattern's CSC order; hess_map: 0-based positions
        into nzval of the Hessian's entries (`_diag_indices` / `_sparse_hessian_map` of the reference)."""
        pv = np.ascontiguousarray(prior_nzval, dtype=np.float64)
        if pv.shape != (self._nnz,):
            raise ValueError(f"prior precision has {pv.size} stored entries but the workspace pattern has {self._nnz}")
        hm = np.ascontiguousarray(hess_map, dtype=np.int64)
        self._hess_cnt = int(hm.size)
        check(lib().gmrfx_set_prior(self._h, ptr(pv), ptr(hm), hm.size, 0), self._h)

    def refactorize_update(self, hvals) -> int:
        """Q <- Q_prior - H (entries at hess_map), refactorise; only `hvals` crosses PCIe."""
        hv = np.ascontiguousarray(hvals, dtype=np.float64)
        if hv.size != getattr(self, "_hess_cnt", -1):
            raise ValueError("Hessian values do not match the index map passed to set_prior")
        info = C.c_int64(0)
        check(lib().gmrfx_refactorize_update(self._h, ptr(hv), C.byref(info)), self._h)
        self._selinv_cache = None
        self._selinv_diag_cache = None
        self.last_info = info.value
        return info.value

    def refactorize_update_solve(self, hvals, rhs):
        """One Newton iterate in one pipelined call: Q <- Q_prior - H, refactorise, solve Q X = rhs (gmrfx_refactorize_update_solve)."""
        hv = np.ascontiguousarray(hvals, dtype=np.float64)
        if hv.size != getattr(self, "_hess_cnt", -1):
            raise ValueError("Hessian values do not match the index map passed to set_prior")
        B = np.asarray(rhs, dtype=np.float64)
        if B.shape[0] != self.n:
            raise ValueError("dimension mismatch")
        vec = B.ndim == 1
        Bf = np.asfortranarray(B.reshape(self.n, -1))
        X = np.empty_like(Bf, order="F")
        info = C.c_int64(0)
        check(lib().gmrfx_refactorize_update_solve(self._h, ptr(hv), ptr(Bf), self.n, Bf.shape[1], ptr(X), self.n, C.byref(info)), self._h)
        self._selinv_cache = None
        self._selinv_diag_cache = None
        self.last_info = info.value
        return X[:, 0].copy() if vec else X

    # -- sharded factorisation (include/gmrfx.h "sharded factorisation"; driver: gmrfx/shard.py) -------
    def refactorize_phase_dev(self, d_nzval_ptr: int, phase: int) -> None:
        check(lib().gmrfx_refactorize_phase(self._h, d_nzval_ptr, phase), self._h)

    def shard_info(self) -> dict:
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().gmrfx_shard_info(self._h, C.byref(a), C.byref(b), C.byref(c)), self._h)
        st = self.stats()
        return {"n_edges": a.value, "n_top_fronts": b.value, "shard_level": c.value, "n_top_levels": int(st["nlevels"] - c.value)}

    def shard_edges(self) -> dict:
        """Cross-rank tree edges child -> parent, ordered by the level of the parent: child, src (owner of the
        child), dst (owner of the parent), level, and where the child's contribution block / update vector live
        (offset + count in doubles into device_ptr(0); first row + rows of device_ptr(3))."""
        k = self.shard_info()["n_edges"]
        names = ("child", "src", "dst", "level", "cb_offset", "cb_count", "w_row0", "w_nrows", "zb_offset", "child_level")
        arr = {nm: np.zeros(k, np.int64) for nm in names}
        if k:
            check(lib().gmrfx_shard_edges(self._h, *[ptr(arr[nm]) for nm in names]), self._h)
        return arr

    def selinv_phase(self, what: int, hi: int = 0, lo: int = 0) -> None:
        """Sharded selected inversion (include/gmrfx.h): 0 begin, 1 gather for other ranks at level hi, 2 own levels hi-1..lo, 3 end."""
        self._selinv_cache = None
        self._selinv_diag_cache = None
        check(lib().gmrfx_selinv_phase(self._h, what, hi, lo), self._h)

    def shard_owner(self, with_top: bool = False):
        ns = self.stats()["nsuper"]
        out, top = np.zeros(ns, np.int64), np.zeros(ns, np.int64)
        check(lib().gmrfx_shard_owner(self._h, ptr(out), ptr(top)), self._h)
        return (out, top.astype(bool)) if with_top else out

    def solve_phase_dev(self, d_B: int, ldb: int, nrhs: int, d_X: int, ldx: int, phase: int) -> None:
        check(lib().gmrfx_solve_phase(self._h, d_B, ldb, nrhs, d_X, ldx, phase), self._h)

    def shard_rows(self, kind: int):
        """(owner, first row, rows, level) of the X row blocks the sharded solve moves (2: own columns of the top
        fronts, broadcast by their owners; 3: columns of the assigned subtrees, gathered on rank 0)."""
        k = C.c_int64(0)
        check(lib().gmrfx_shard_rows(self._h, kind, C.byref(k), None, None, None, None), self._h)
        owner, r0, nr, lv = (np.zeros(k.value, np.int64) for _ in range(4))
        if k.value:
            check(lib().gmrfx_shard_rows(self._h, kind, C.byref(k), ptr(owner), ptr(r0), ptr(nr), ptr(lv)), self._h)
        return owner, r0, nr, lv

    def shard_dist_fronts(self) -> dict:
        """The distributed top fronts of a sharded handle (csrc/symbolic.h: Symbolic::dist_fronts): per front its supernode,
        columns, rows, the panel's place in device_ptr(1), tree level and GROUP (the ranks that factor it together: panel block b
        of 256 columns on group[b % g], contribution-block block q on group[(panel blocks + q) % g])."""
        cnt = np.zeros(4, np.int64)
        check(lib().gmrfx_shard_dist_fronts(self._h, ptr(cnt), *([None] * 8)))
        nf, ng = int(cnt[0]), int(cnt[1])
        a = {nm: np.empty(nf, np.int64) for nm in ("front", "cols", "rows", "panel_offset", "panel_ld", "level")}
        gptr, grank = np.zeros(nf + 1, np.int64), np.empty(ng, np.int64)
        check(lib().gmrfx_shard_dist_fronts(self._h, ptr(cnt), *[ptr(a[nm]) for nm in ("front", "cols", "rows", "panel_offset", "panel_ld", "level")],
                                            ptr(gptr), ptr(grank)))
        a["group"] = [[int(r) for r in grank[gptr[k]:gptr[k + 1]]] for k in range(nf)]
        a["world"] = int(cnt[3])
        return a

    def shard_transfers(self) -> dict:
        """Every contribution-block transfer of the sharded factorisation (whole columns of `child`'s block: `count` doubles at
        `offset` of device_ptr(0), src -> dst, before the fronts of `level` are assembled), ordered by level."""
        cnt = np.zeros(4, np.int64)
        check(lib().gmrfx_shard_dist_fronts(self._h, ptr(cnt), *([None] * 8)))
        k = int(cnt[2])
        names = ("child", "src", "dst", "level", "offset", "count", "col0")
        a = {nm: np.empty(k, np.int64) for nm in names}
        check(lib().gmrfx_shard_transfers(self._h, *[ptr(a[nm]) for nm in names]))
        return a

    def dist_front_phase(self, d_nzval_ptr: int, front: int, what: int, block: int = 0) -> None:
        check(lib().gmrfx_dist_front_phase(self._h, d_nzval_ptr, front, what, block), self._h)

    def set_stream(self, hip_stream: int, use_external: bool = True, async_phases: bool = False) -> None:
        """The caller's HIP stream becomes the handle's main stream (sharded drivers: torch's current stream)."""
        check(lib().gmrfx_set_stream(self._h, C.c_void_p(int(hip_stream)), int(use_external), int(async_phases)), self._h)

    def level_times(self, which: int) -> np.ndarray:
        """ms per tree level of the last factorisation (0) / forward (1) / backward (2) sweep ([0] = the sweep tasks); empty
        unless the handle was created under GMRFX_LEVEL_MARK=1."""
        out = np.zeros(int(self.stats()["nlevels"]) + 1)
        cnt = C.c_int64(0)
        check(lib().gmrfx_level_times(self._h, which, ptr(out), out.size, C.byref(cnt)), self._h)
        return out[:cnt.value]

    def dist_front_block(self, front: int, block: int):
        """(offset, count) in doubles of THIS rank's copy of panel block `block` of the distributed front `front` inside
        device_ptr(1): the whole panel on the front's owner, own blocks + a two-block window on a member with block-cyclic storage"""
        off, cnt = C.c_int64(0), C.c_int64(0)
        check(lib().gmrfx_dist_front_block(self._h, int(front), int(block), C.byref(off), C.byref(cnt)), self._h)
        return off.value, cnt.value

    def device_ptr(self, which: int) -> int:
        return int(lib().gmrfx_device_ptr(self._h, which) or 0)

    def logdet_partial(self) -> float:
        out = C.c_double(0.0)
        check(lib().gmrfx_logdet_partial(self._h, C.byref(out)), self._h)
        return out.value

    def solve_dev(self, d_B: int, ldb: int, nrhs: int, d_X: int, ldx: int) -> None:
        check(lib().gmrfx_solve_dev(self._h, d_B, ldb, nrhs, d_X, ldx), self._h)

    def backward_solve_dev(self, d_Z: int, ldz: int, nrhs: int, d_X: int, ldx: int) -> None:
        check(lib().gmrfx_backward_solve_dev(self._h, d_Z, ldz, nrhs, d_X, ldx), self._h)

    def quadform_dev(self, d_nzval: int, d_X: int, ldx: int, nvec: int, d_mu: int = 0) -> np.ndarray:
        """Device-pointer form of sqmahal; d_nzval / d_mu may be 0 (see gmrfx_quadform_dev)."""
        out = np.empty(nvec)
        check(lib().gmrfx_quadform_dev(self._h, d_nzval or None, d_X, ldx, nvec, d_mu or None, ptr(out)), self._h)
        return out

    def selinv_compute_dev(self) -> None:
        check(lib().gmrfx_selinv_compute(self._h), self._h)
