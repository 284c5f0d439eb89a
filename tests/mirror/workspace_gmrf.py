"""TEST INFRASTRUCTURE: `WorkspaceGMRF` + `ConstraintInfo` of src/workspace/workspace_gmrf.jl:22-56, 230-305 on the
GMRFWorkspace mirror. The only calls into the backend are the ones the reference makes: ONE blocked n x m multi-RHS
`workspace_solve` for A~' = Q^-1 A' (:37), `selinv_diag`, `logdet`, `backward_solve`; the m x m Cholesky and the
corrections are host arithmetic exactly as written there."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .workspace import GMRFWorkspace


class ConstraintInfo:
    def __init__(self, ws: GMRFWorkspace, mu, A, e):
        n = ws.dimension()
        A_sp = sp.csc_matrix(A)
        m, n_A = A_sp.shape
        if n != n_A:
            raise ValueError(f"Constraint matrix size {A_sp.shape} incompatible with workspace size {n}")
        e = np.asarray(e, dtype=np.float64)
        if m != e.shape[0]:
            raise ValueError(f"Constraint matrix rows {m} != constraint vector length {e.shape[0]}")
        self.matrix, self.vector = A_sp, e
        # A~' = Q^-1 A' via one blocked multi-RHS solve (workspace_gmrf.jl:37)
        self.A_tilde_T = ws.workspace_solve(np.asfortranarray(A_sp.T.toarray())).reshape(n, m)
        self.L_c = np.linalg.cholesky(np.asarray(A_sp @ self.A_tilde_T))
        mu = np.asarray(mu, dtype=np.float64)
        self.constrained_mean = mu - self.A_tilde_T @ self._lc_solve(A_sp @ mu - e)
        resid_e = e - A_sp @ mu
        logdet_Lc = 2.0 * np.log(np.diag(self.L_c)).sum()
        AAt = np.asarray((A_sp @ A_sp.T).todense())
        self.log_constraint_correction = 0.5 * (m * np.log(2.0 * np.pi) + logdet_Lc + resid_e @ self._lc_solve(resid_e)) \
            - 0.5 * np.linalg.slogdet(AAt)[1]

    def _lc_solve(self, v):
        return np.linalg.solve(self.L_c.T, np.linalg.solve(self.L_c, v))


class WorkspaceGMRF:
    def __init__(self, mean, precision, workspace: GMRFWorkspace, A=None, e=None):
        self.mean_ = np.asarray(mean, dtype=np.float64)
        self.precision = sp.csc_matrix(precision)
        self.workspace = workspace
        workspace.update_precision(self.precision)
        self.version = workspace.next_version
        workspace.next_version += 1
        workspace.loaded_version = self.version
        self.constraints = None if A is None else ConstraintInfo(workspace, self.mean_, A, e)

    def ensure_loaded(self):
        ws = self.workspace
        if ws.loaded_version != self.version:
            ws.update_precision_values(self.precision.data)
            ws.loaded_version = self.version

    def mean(self):
        return self.mean_ if self.constraints is None else self.constraints.constrained_mean

    def logdetcov(self) -> float:
        self.ensure_loaded()
        return self.workspace.logdet_cov()

    def var(self):
        self.ensure_loaded()
        sigma = self.workspace.selinv_diag()
        if self.constraints is None:
            return sigma
        ci = self.constraints
        B_T = np.linalg.solve(ci.L_c, ci.A_tilde_T.T)               # L_c.L \ A~'   (m x n)
        return np.maximum(sigma - (B_T ** 2).sum(axis=0), 0.0)

    def rand(self, rng: np.random.Generator, k: int):
        self.ensure_loaded()
        n = self.precision.shape[0]
        X = self.workspace.backward_solve(rng.standard_normal((n, k))).reshape(n, k) + self.mean_[:, None]
        if self.constraints is not None:
            ci = self.constraints
            X -= ci.A_tilde_T @ ci._lc_solve(ci.matrix @ X - ci.vector[:, None])
        return X

    def rand_from(self, Z):
        """The batched sampler of julia/GMRFX.jl (Distributions._rand!(rng, d::WorkspaceGMRF{<:Any, MI355XBackend}, X::AbstractMatrix))
        on given standard-normal draws Z (n x k): ONE multi-column backward sweep, mean and constraint correction on all columns."""
        self.ensure_loaded()
        Z = np.asarray(Z, dtype=np.float64)
        X = self.workspace.backward_solve(Z).reshape(Z.shape) + self.mean_[:, None]
        if self.constraints is not None:
            ci = self.constraints
            X -= ci.A_tilde_T @ ci._lc_solve(ci.matrix @ X - ci.vector[:, None])
        return X

    def rand_from_column_by_column(self, Z):
        """What the reference does with the same draws (workspace_gmrf.jl:275-286 under Distributions' column loop): one
        single-RHS backward sweep, one mean shift and one constraint correction per sample."""
        self.ensure_loaded()
        Z = np.asarray(Z, dtype=np.float64)
        X = np.empty_like(Z)
        for j in range(Z.shape[1]):
            x = self.workspace.backward_solve(Z[:, j].copy()) + self.mean_
            if self.constraints is not None:
                ci = self.constraints
                x = x - ci.A_tilde_T @ ci._lc_solve((ci.matrix @ x - ci.vector)[:, None])[:, 0]
            X[:, j] = x
        return X

    def logpdf(self, z) -> float:
        self.ensure_loaded()
        r = np.asarray(z) - self.mean_
        n = r.shape[0]
        val = -0.5 * float(r @ (self.precision @ r)) - 0.5 * self.logdetcov() - 0.5 * n * np.log(2.0 * np.pi)
        if self.constraints is not None:
            val += self.constraints.log_constraint_correction
        return val
