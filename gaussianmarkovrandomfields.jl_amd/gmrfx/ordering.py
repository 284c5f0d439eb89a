"""`ordering` keyword of the backend constructor -- host logic of src/workspace/backend.jl:73-133.

The reference accepts `nothing` | a permutation vector | a CliqueTrees elimination ALGORITHM object |
`PinDenseColumns(inner; frac)`; `ordering_permutation(A, ordering)` resolves any of them to an explicit permutation
ONCE, so that every consumer of the pattern (all members of a WorkspacePool) shares it. Here:

  None            libgmrfx's own nested dissection (geometric when coords are given)
  "natural"       identity
  array           explicit permutation (0-based)
  callable        stands where a CliqueTrees algorithm object stands in Julia: `alg(pattern_csc) -> permutation`
  PinDenseColumns columns with more than frac * n stored entries are pinned to the END of the elimination order,
                  `inner` (any of the above) orders the remaining sparse block

Anything else raises TypeError -- an ordering request is never dropped silently."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


class PinDenseColumns:
    """PinDenseColumns(inner; frac = 0.5), src/workspace/backend.jl:73-77."""

    def __init__(self, inner=None, frac: float = 0.5):
        self.inner, self.frac = inner, float(frac)


def _check_perm(p, n):
    p = np.ascontiguousarray(p, dtype=np.int64).reshape(-1)
    if p.shape != (n,) or not np.array_equal(np.sort(p), np.arange(n)):
        raise ValueError("ordering must be a permutation of 0..n-1")
    return p


def ordering_permutation(A, ordering, coords=None):
    """Explicit permutation (0-based int64) for A's pattern, or None when libgmrfx should run its own nested
    dissection on the whole pattern / "natural" should be passed through (returned as the string)."""
    A = sp.csc_matrix(A)
    n = A.shape[1]
    if ordering is None:
        return None
    if isinstance(ordering, str):
        if ordering != "natural":
            raise ValueError(f"unknown ordering {ordering!r}")
        return "natural"
    if isinstance(ordering, PinDenseColumns):
        cnt = np.diff(A.indptr)
        dense = np.flatnonzero(cnt > ordering.frac * n)
        if dense.size == 0:                       # transparent passthrough (backend.jl:100)
            return ordering_permutation(A, ordering.inner, coords)
        keep = np.flatnonzero(cnt <= ordering.frac * n)
        S = sp.csc_matrix(A[keep][:, keep])       # pattern of the sparse block (values are irrelevant)
        inner = ordering_permutation(S, ordering.inner, None if coords is None else np.asarray(coords)[keep])
        if inner is None:                         # own nested dissection on the sparse block
            from .backend import MI355XBackend
            S.data = np.ones_like(S.data, dtype=np.float64)
            sub = MI355XBackend(S, coords=None if coords is None else np.asarray(coords)[keep], symbolic_only=True)
            inner = sub.ordering_permutation()
            sub.close()
        elif isinstance(inner, str):
            inner = np.arange(keep.size)
        return np.concatenate([keep[inner], dense]).astype(np.int64)
    if callable(ordering):
        P = A.copy()
        P.data = np.ones_like(P.data, dtype=np.float64)
        return _check_perm(ordering(P), n)
    if isinstance(ordering, (np.ndarray, list, tuple)):
        return _check_perm(ordering, n)
    raise TypeError(f"unsupported ordering specification {type(ordering).__name__}: expected None, 'natural', a permutation, "
                    "a callable pattern -> permutation, or PinDenseColumns")
