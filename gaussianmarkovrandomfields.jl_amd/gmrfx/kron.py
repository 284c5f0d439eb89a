"""Separable (Kronecker) precision Q = kron(Q_1, Q_2) -- SURVEY 8 f3, prior-level operations only: everything is
answered from the two factor-scale backends, the n_1 n_2 x n_1 n_2 product is never formed.

Reference: SeparableModel, src/latent_models/separable.jl:122-172 -- `precision_matrix` folds `kron` over the
components (the RIGHTMOST component varies fastest: x[i1 * n2 + i2]), `precision_logdet` uses
logdet(Q_1 (x) Q_2) = n_2 logdet(Q_1) + n_1 logdet(Q_2). With X = reshape(x, n2, n1) (column i1 = component-2
vector at index i1 of component 1): (Q_1 (x) Q_2) x = vec(Q_2 X Q_1'), so solves, samples and marginal variances
are two batched sweeps -- n_1 right-hand sides on factor 2, then n_2 on factor 1."""
from __future__ import annotations

import numpy as np

from .backend import MI355XBackend


class KroneckerWorkspace:
    def __init__(self, Q1, Q2, kw1=None, kw2=None):
        self.ws1 = MI355XBackend(Q1, **(kw1 or {}))
        self.ws2 = MI355XBackend(Q2, **(kw2 or {}))
        self.n1, self.n2 = self.ws1.n, self.ws2.n

    def dimension(self) -> int:
        return self.n1 * self.n2

    def _mat(self, x):
        x = np.asarray(x, dtype=np.float64)
        if x.shape != (self.n1 * self.n2,):
            raise ValueError("dimension mismatch")
        return np.asfortranarray(x.reshape(self.n1, self.n2).T)      # n2 x n1

    @staticmethod
    def _vec(Xm):
        return np.ascontiguousarray(Xm.T).reshape(-1)

    def logdet(self) -> float:
        """precision_logdet (separable.jl:122-141): sum_i (N / n_i) logdet(Q_i)."""
        return self.n2 * self.ws1.compute_logdet() + self.n1 * self.ws2.compute_logdet()

    def solve(self, b):
        """(Q_1 (x) Q_2)^-1 b = vec(Q_2^-1 B Q_1^-1)."""
        W = self.ws2.backend_solve(self._mat(b)).reshape(self.n2, self.n1)
        Y = self.ws1.backend_solve(np.asfortranarray(W.T)).reshape(self.n1, self.n2)
        return self._vec(Y.T)

    def backward_solve(self, z):
        """A sample of N(0, Q^-1) from z ~ N(0, I): (A_1 (x) A_2) z with A_i = P_i' L_i^-T (backend.jl:281-284)."""
        W = self.ws2.backend_backward_solve(self._mat(z)).reshape(self.n2, self.n1)
        Y = self.ws1.backend_backward_solve(np.asfortranarray(W.T)).reshape(self.n1, self.n2)
        return self._vec(Y.T)

    def selinv_diag(self):
        """diag(Q^-1) = kron(diag(Q_1^-1), diag(Q_2^-1))."""
        return np.kron(self.ws1.get_selinv_diag(), self.ws2.get_selinv_diag())
