"""Host-side mirror of the reference's KL (Vecchia) sparse approximate Cholesky entry points
(src/kl_cholesky/kl_cholesky.jl) over gmrfx_kl_cholesky: the batch of small dense factorisations runs on the
MI355X, the sparsity pattern (reverse maximin ordering, rho-neighbourhoods, supernode clustering) is the caller's."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from ._lib import check, lib, ptr


def _run(Theta, L_colptr, task_rowptr, task_rows, task_colptr, task_cols, reg, device, theta_device_ptr=None, n=None):
    if theta_device_ptr is None:
        Th = np.asfortranarray(Theta, dtype=np.float64)
        n = Th.shape[0]
        if Th.shape != (n, n):
            raise ValueError("Theta must be square")
        th_ptr, on_dev = Th.ctypes.data, 0
    else:
        th_ptr, on_dev = theta_device_ptr, 1
    L_colptr = np.ascontiguousarray(L_colptr, dtype=np.int64)
    arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (task_rowptr, task_rows, task_colptr, task_cols)]
    nz = np.empty(int(L_colptr[-1]))
    info = C.c_int64(0)
    check(lib().gmrfx_kl_cholesky(n, th_ptr, n, on_dev, ptr(L_colptr), len(arrs[0]) - 1, ptr(arrs[0]), ptr(arrs[1]),
                                  ptr(arrs[2]), ptr(arrs[3]), 0, float(reg), device, ptr(nz), C.byref(info)))
    return nz


def sparse_approximate_cholesky_inplace(Theta, L: sp.csc_matrix, reg: float = 1e-6, device: int = -1,
                                        theta_device_ptr=None) -> sp.csc_matrix:
    """sparse_approximate_cholesky!(Theta, L) (kl_cholesky.jl:32-55): fills the values of the lower-triangular
    pattern L so that L L' ~ Theta^-1. Returns a new csc_matrix with L's pattern (scipy arrays are not Julia's)."""
    L = sp.csc_matrix(L)
    L.sort_indices()
    n = L.shape[0]
    colptr = L.indptr.astype(np.int64)
    # one task per column: its rows in descending order
    rows = np.concatenate([L.indices[colptr[k]:colptr[k + 1]][::-1] for k in range(n)]) if n else np.zeros(0, np.int64)
    nz = _run(Theta, colptr, colptr, rows, np.arange(n + 1), np.arange(n), reg, device, theta_device_ptr, n)
    return sp.csc_matrix((nz, L.indices.copy(), L.indptr.copy()), shape=L.shape)


def supernodal_pattern(column_indices, row_indices, n: int) -> sp.csc_matrix:
    """_build_supernodal_sparsity_pattern (kl_cholesky.jl:57-72): entry (i, j) for every member column j and
    every row i >= j of its supernode."""
    Is, Js = [], []
    for cols, rows in zip(column_indices, row_indices):
        for j in cols:
            for i in rows:
                if j <= i:
                    Is.append(i); Js.append(j)
    P = sp.csc_matrix((np.ones(len(Is)), (Is, Js)), shape=(n, n))
    P.sum_duplicates()
    P.sort_indices()
    return P


def sparse_approximate_cholesky_supernodal(Theta, column_indices, row_indices, reg: float = 1e-8, device: int = -1,
                                           theta_device_ptr=None, n=None) -> sp.csc_matrix:
    """sparse_approximate_cholesky(Theta, sc::SupernodeClustering) (kl_cholesky.jl:74-113). row_indices[s] must be
    in DESCENDING order (the reference keeps them in a SortedSet(Base.Reverse), supernodes.jl:71)."""
    if n is None:
        n = np.asarray(Theta).shape[0]
    P = supernodal_pattern(column_indices, row_indices, n)
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in row_indices])])
    colptr_t = np.concatenate([[0], np.cumsum([len(c) for c in column_indices])])
    rows = np.concatenate([np.asarray(r, dtype=np.int64) for r in row_indices])
    cols = np.concatenate([np.asarray(c, dtype=np.int64) for c in column_indices])
    nz = _run(Theta, P.indptr.astype(np.int64), rowptr, rows, colptr_t, cols, reg, device, theta_device_ptr, n)
    return sp.csc_matrix((nz, P.indices.copy(), P.indptr.copy()), shape=P.shape)


def radius_pattern(X: np.ndarray, rho_len: float) -> sp.csc_matrix:
    """A simple lower-triangular test pattern for points X (n x d, already in elimination order): (i, k), i >= k,
    whenever |x_i - x_k| <= rho_len. (The reference derives its pattern from the reverse maximin ordering,
    maximin.jl; any lower-triangular pattern with a full diagonal is valid input for the kernels.)"""
    from scipy.spatial import cKDTree
    t = cKDTree(X)
    pairs = t.query_pairs(rho_len, output_type="ndarray")
    n = X.shape[0]
    i = np.maximum(pairs[:, 0], pairs[:, 1]); k = np.minimum(pairs[:, 0], pairs[:, 1])
    P = sp.csc_matrix((np.ones(len(i) + n), (np.concatenate([i, np.arange(n)]), np.concatenate([k, np.arange(n)]))), shape=(n, n))
    P.sort_indices()
    return P
