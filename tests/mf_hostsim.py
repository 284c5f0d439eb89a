"""Test helper (CPU, numpy): walks the symbolic structure exported by libgmrfx.so
(gmrfx_symbolic_get) and performs the same multifrontal factorisation / sweeps / Takahashi
recursion the HIP kernels perform, with dense numpy blocks. It validates the HOST logic (scatter
map, relative indices, level order) on machines without a GPU. Test infrastructure only."""
import numpy as np


class HostSim:
    def __init__(self, sy, n, nzval):
        self.sy, self.n = sy, n
        ns = len(sy.super_parent)
        self.ns = ns
        self.c = np.diff(sy.super_first)
        self.r = np.diff(sy.row_ptr)
        self.L = np.zeros(int(sy.panel_ptr[-1]))
        here = sy.q_dst >= 0            # (-1: the entry belongs to a panel another rank of a sharded factorisation stores)
        self.L[sy.q_dst[here]] = nzval[sy.q_src[here]]
        self.children = [[] for _ in range(ns)]
        for s in range(ns):
            if sy.super_parent[s] >= 0:
                self.children[sy.super_parent[s]].append(s)
        self.order = np.lexsort((np.arange(ns), sy.level))   # level order, as on the device

    def rows(self, s):
        return self.sy.rows[self.sy.row_ptr[s]:self.sy.row_ptr[s + 1]]

    def rel(self, s):
        return self.sy.rel[self.sy.row_ptr[s] + self.c[s]:self.sy.row_ptr[s + 1]]

    def panel(self, arr, s):
        ld, c, r = self.sy.panel_ld[s], self.c[s], self.r[s]
        p = self.sy.panel_ptr[s]
        return arr[p:p + ld * c].reshape(c, ld).T[:r]      # view r x c (column-major storage)

    def factor(self):
        cb = {}
        for s in self.order:
            c, r = self.c[s], self.r[s]
            F = np.zeros((r, r))
            P = self.panel(self.L, s)
            F[:, :c] = P
            for d in self.children[s]:
                rel = self.rel(d)
                F[np.ix_(rel, rel)] += cb.pop(d)
            F = np.tril(F)
            F = F + np.tril(F, -1).T
            L11 = np.linalg.cholesky(F[:c, :c])
            L21 = np.linalg.solve(L11, F[c:, :c].T).T
            P[:c, :] = np.tril(L11)
            P[c:, :] = L21
            cb[s] = F[c:, c:] - L21 @ L21.T
        return self

    def solve(self, Bp, mode=0):
        """Bp: n x k in elimination order; mode 0 full solve, 1 backward only."""
        X = np.array(Bp, dtype=float, copy=True)
        if mode == 0:
            W = {}
            for s in self.order:
                c, r = self.c[s], self.r[s]
                f = np.zeros((r, X.shape[1]))
                rows = self.rows(s)
                f[:c] = X[rows[:c]]
                for d in self.children[s]:
                    f[self.rel(d)] += W.pop(d)
                P = self.panel(self.L, s)
                y = np.linalg.solve(np.tril(P[:c]), f[:c])
                X[rows[:c]] = y
                W[s] = f[c:] - P[c:] @ y
        for s in self.order[::-1]:
            c = self.c[s]
            rows = self.rows(s)
            P = self.panel(self.L, s)
            y = X[rows[:c]] - P[c:].T @ X[rows[c:]]
            X[rows[:c]] = np.linalg.solve(np.tril(P[:c]).T, y)
        return X

    def logdet(self):
        return 2.0 * sum(np.log(np.diag(self.panel(self.L, s)[:self.c[s]])).sum() for s in range(self.ns))

    def selinv(self):
        Z = np.zeros_like(self.L)
        ZB = {}
        for s in self.order[::-1]:
            c, r = self.c[s], self.r[s]
            p = self.sy.super_parent[s]
            P = self.panel(self.L, s)
            Zs = self.panel(Z, s)
            if r > c:
                rel = self.rel(s)
                cp = self.c[p]
                Zp = np.zeros((self.r[p], self.r[p]))
                Zp[:, :cp] = self.panel(Z, p)
                Zp[cp:, cp:] = ZB[p]
                Zp = np.tril(Zp) + np.tril(Zp, -1).T
                Z22 = Zp[np.ix_(rel, rel)]
            else:
                Z22 = np.zeros((0, 0))
            L11 = np.tril(P[:c])
            Y = np.linalg.solve(L11.T, P[c:].T).T          # L21 L11^-1
            Z21 = -Z22 @ Y
            Li = np.linalg.inv(L11)
            Z11 = Li.T @ Li - Y.T @ Z21
            Zs[:c] = Z11
            Zs[c:] = Z21
            ZB[s] = Z22
        return Z

    def dense_from_panels(self, arr):
        """Lower-triangular n x n dense matrix (elimination order) from panel storage."""
        M = np.zeros((self.n, self.n))
        for s in range(self.ns):
            rows = self.rows(s)
            P = self.panel(arr, s)
            for j in range(self.c[s]):
                M[rows[j:], rows[j]] = P[j:, j]
        return M
