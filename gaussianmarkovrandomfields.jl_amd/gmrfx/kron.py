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

    # ---- device-resident path ------------------------------------------------------------------------------------
    # x[i1 * n2 + i2] IS the column-major n2 x n1 matrix X with leading dimension n2, so the sweep over factor 2 takes the
    # flat device vector as it is (n1 right-hand sides, passes of 64 on the device). Its result, seen as the ROW-major
    # n1 x n2 matrix T, is then multiplied from the left by the small factor's DENSE operator D1 (n1 x n1: Q_1^-1 for solves,
    # A_1 = P_1' L_1^-T for samples -- n1 solves of the small factor, once): R = D1 T is already laid out as the answer,
    # R[i1, i2] = out[i1 * n2 + i2]. No host panel, no transpose. D1 T runs on the library's own FP64-MFMA kernel
    # (gmrfx_dense_apply_dev, csrc/dense.hip) -- round 3 used torch.matmul (rocBLAS) here.
    # The dense operator is used when the first factor is small (the time factor of cfg 5: 512); otherwise both factors sweep,
    # with the layout change between them done by gmrfx_transpose_dev. torch only allocates the buffers.
    DENSE_MAX = 4096

    def _dense_op(self, kind: str):
        import torch
        cache = self.__dict__.setdefault("_dense", {})
        if kind not in cache:
            eye = np.asfortranarray(np.eye(self.n1))
            D = self.ws1.backend_solve(eye) if kind == "solve" else self.ws1.backend_backward_solve(eye)
            cache[kind] = torch.from_numpy(np.ascontiguousarray(np.asarray(D).reshape(self.n1, self.n1))).to(self._device())
        return cache[kind]

    def _device(self):
        import torch
        return torch.device("cuda", self.ws2.device if self.ws2.device >= 0 else torch.cuda.current_device())

    @staticmethod
    def _buf(like, count):
        """count doubles on like's device, with 16 bytes of slack behind them (the kernels load operands in 16-byte pairs)"""
        import torch
        return torch.empty(count + 2, dtype=torch.float64, device=like.device)[:count]

    def _apply_dev(self, x, kind: str):
        import torch
        n1, n2 = self.n1, self.n2
        if x.dtype != torch.float64 or x.numel() != n1 * n2 or not x.is_contiguous() or not x.is_cuda:
            raise ValueError("expected a contiguous float64 CUDA tensor with n1 * n2 entries")
        w = self._buf(x, n1 * n2)
        torch.cuda.synchronize(x.device)            # torch's stream -> the library's own streams
        f2 = self.ws2.solve_dev if kind == "solve" else self.ws2.backward_solve_dev
        f2(x.data_ptr(), n2, n1, w.data_ptr(), n2)                       # factor 2 on all n1 columns (returns synchronised)
        if n1 <= self.DENSE_MAX:
            D = self._dense_op(kind)
            out = self._buf(x, n1 * n2)
            torch.cuda.synchronize(x.device)
            self.ws2.dense_apply_dev(D.data_ptr(), n1, w.data_ptr(), n2, out.data_ptr())       # R = D1 T, row-major
            return out
        wt = self._buf(x, n1 * n2)                                      # column-major n1 x n2 (= row-major n2 x n1), leading dimension n1
        y = self._buf(x, n1 * n2)
        torch.cuda.synchronize(x.device)
        self.ws2.transpose_dev(w.data_ptr(), n1, n2, wt.data_ptr())
        f1 = self.ws1.solve_dev if kind == "solve" else self.ws1.backward_solve_dev
        f1(wt.data_ptr(), n1, n2, y.data_ptr(), n1)
        self.ws2.transpose_dev(y.data_ptr(), n2, n1, w.data_ptr())       # back to row-major n1 x n2 = the flat answer
        return w

    def solve_dev(self, x):
        """(Q_1 (x) Q_2)^-1 x for a device-resident x (torch CUDA tensor, float64, n1 n2 entries); returns a new tensor."""
        return self._apply_dev(x, "solve")

    def backward_solve_dev(self, z):
        """A sample of N(0, Q^-1) from a device-resident z ~ N(0, I)."""
        return self._apply_dev(z, "sample")
