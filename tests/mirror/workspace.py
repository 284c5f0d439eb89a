"""TEST INFRASTRUCTURE (not part of the product package): GMRFWorkspace / WorkspacePool -- mirror of src/workspace/gmrf_workspace.jl:31-302 and
src/workspace/workspace_pool.jl:42-119 on top of MI355XBackend: owns a Q buffer with a fixed
pattern, lazy validity flags, the logdet cache; every numeric result comes from the backend."""
from __future__ import annotations

import queue
from contextlib import contextmanager

import numpy as np
import scipy.sparse as sp
from gmrfx._lib import PosDefException

from gmrfx.backend import MI355XBackend, _as_csc


class GMRFWorkspace:
    def __init__(self, Q, backend_type=MI355XBackend, **backend_kwargs):
        Q = _as_csc(Q)
        self.Q = Q.copy()
        self.backend = backend_type(Q, **backend_kwargs)
        self.numeric_valid = True      # ctor factorizes (gmrf_workspace.jl:72-84)
        self.selinv_valid = False
        self.logdet_valid = False
        self.logdet_cache = 0.0
        self.next_version = 1
        self.loaded_version = 0

    def dimension(self) -> int:
        return self.Q.shape[0]

    def _invalidate(self):
        self.numeric_valid = False
        self.selinv_valid = False
        self.logdet_valid = False

    def update_precision(self, Q_new) -> None:
        Q_new = _as_csc(Q_new)
        same = (Q_new.shape == self.Q.shape and np.array_equal(Q_new.indptr, self.Q.indptr)
                and np.array_equal(Q_new.indices, self.Q.indices))
        if not same:
            raise ValueError("Sparsity pattern mismatch: Q_new has different colptr/rowval. "
                             "GMRFWorkspace requires the same sparsity pattern across updates.")
        self.Q.data[:] = Q_new.data
        self._invalidate()
        self.loaded_version = 0

    def update_precision_values(self, nzval) -> None:
        nzval = np.asarray(nzval, dtype=np.float64)
        if nzval.shape[0] != self.Q.data.shape[0]:
            raise ValueError(f"nzval length {nzval.shape[0]} does not match workspace Q nzval length {self.Q.data.shape[0]}")
        self.Q.data[:] = nzval
        self._invalidate()
        self.loaded_version = 0

    def ensure_numeric(self) -> None:
        if not self.numeric_valid:
            self.backend.refactorize(self.Q)
            self.numeric_valid = True
            self.selinv_valid = False
            self.logdet_valid = False

    def ensure_selinv(self) -> None:
        if not self.selinv_valid:
            self.ensure_numeric()
            self.backend.compute_selinv()
            self.selinv_valid = True

    def workspace_solve(self, b):
        # gmrf_workspace.jl:207-215. With stale values the reference's ensure_numeric! + backend_solve become the backend's ONE
        # pipelined call (gmrfx_refactorize_solve, julia/GMRFX.jl refactorize_solve!); same bits as the two calls.
        if (not self.numeric_valid and hasattr(self.backend, "refactorize_solve") and sp.isspmatrix_csc(self.Q)
                and self.Q.has_sorted_indices):
            if self.Q.nnz != self.backend._nnz:
                raise ValueError("the sparsity pattern must be invariant across refactorizations")
            x = self.backend.refactorize_solve(self.Q.data, b)
            if self.backend.last_info > 0:        # `b.factor \ rhs` throws on a failed factor (backend.jl:178-193): never a silent NaN solution
                raise PosDefException(5, f"matrix is not positive definite; Cholesky factorization failed at pivot {self.backend.last_info}")
            self.numeric_valid = True
            self.selinv_valid = False
            self.logdet_valid = False
            return x
        self.ensure_numeric()
        return self.backend.backend_solve(b)

    def logdet(self) -> float:
        if not self.logdet_valid:
            self.ensure_numeric()
            self.logdet_cache = self.backend.compute_logdet()
            self.logdet_valid = True
        return self.logdet_cache

    def logdet_cov(self) -> float:
        return -self.logdet()

    def selinv(self) -> sp.csc_matrix:
        self.ensure_selinv()
        return self.backend.get_selinv()

    def selinv_diag(self) -> np.ndarray:
        self.ensure_selinv()
        return self.backend.get_selinv_diag()

    def selinv_dot(self, B) -> float:
        self.ensure_selinv()
        return self.backend.selinv_dot(B)

    def selinv_extract_at(self, B) -> sp.csc_matrix:
        self.ensure_selinv()
        return self.backend.selinv_extract_at(B)

    def row_diag_ASigmaAt(self, A) -> np.ndarray:
        """diag(A Sigma A') of a sparse design matrix (linear_predictor_marginals.jl:125-165), on the device."""
        self.ensure_selinv()
        return self.backend.row_diag_ASigmaAt(A)

    def backward_solve(self, x):
        self.ensure_numeric()
        return self.backend.backend_backward_solve(x)

    def sqmahal(self, x, mean=None):
        """(x - mean)' Q (x - mean) on the device (gmrf.jl:94-97); x: vector or n x k matrix."""
        self.ensure_numeric()
        return self.backend.sqmahal(x, mean)

    def logpdf(self, z, mean=None) -> float:
        """logpdf(d::WorkspaceGMRF, z) without constraints (workspace_gmrf.jl:288-292)."""
        n = self.dimension()
        return -0.5 * self.sqmahal(z, mean) + 0.5 * self.logdet() - 0.5 * n * np.log(2.0 * np.pi)


class WorkspacePool:
    """N independent workspaces handed out through a queue (workspace_pool.jl:42-119). Distinct
    handles own distinct HIP streams, so checkouts may be used from different threads."""

    def __init__(self, Q, size: int = 2, **kw):
        first = GMRFWorkspace(Q, **kw)
        perm = first.backend.ordering_permutation()
        kw2 = dict(kw)
        kw2["ordering"] = perm        # resolve the ordering once, share it (backend.jl:86-88)
        kw2.pop("coords", None)
        self._all = [first] + [GMRFWorkspace(Q, **kw2) for _ in range(size - 1)]
        self._q = queue.Queue()
        for w in self._all:
            self._q.put(w)

    def checkout(self) -> GMRFWorkspace:
        return self._q.get()

    def checkin(self, ws: GMRFWorkspace) -> None:
        self._q.put(ws)

    @contextmanager
    def with_workspace(self):
        ws = self.checkout()
        try:
            yield ws
        finally:
            self.checkin(ws)

    def __len__(self):
        return len(self._all)
