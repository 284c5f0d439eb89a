"""`ordering` keyword forms of the backend constructor (src/workspace/backend.jl:73-153;
test/workspace/test_backend_ordering.jl:33-54). Host logic only: runs without a GPU on symbolic-only handles;
the GPU half (answers do not depend on the ordering) is in test_seam_a_and_constraints.py."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import gmrfx
import orc
from gmrfx import PinDenseColumns, ordering_permutation
from test_reference_inputs import backend_ordering_matrix


def mmd(P):
    """Stand-in for `CliqueTrees.MMD()`: SuperLU's multiple-minimum-degree ordering of A' + A (a real MMD, an
    independent implementation) obtained from a throw-away LU of the pattern matrix made diagonally dominant."""
    n = P.shape[0]
    M = sp.csc_matrix(P + sp.identity(n) * (2.0 * n))
    return np.asarray(spla.splu(M, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0).perm_c, dtype=np.int64).argsort()


def isperm(p, n):
    return np.array_equal(np.sort(np.asarray(p)), np.arange(n))


def same_elimination(be, Q, p):
    """The backend postorders the elimination tree of the ordering it is given (as CHOLMOD does: F.p is the
    postordered permutation), which relabels pivots without changing the fill: the permutation it reports is a
    permutation, and nnz(L) is exactly that of the requested ordering (oracle, same Q, the requested p)."""
    q = be.ordering_permutation()
    return isperm(q, Q.shape[0]) and be.stats()["nnz_l"] == orc.OracleFactor(Q, p).nnz_L == orc.OracleFactor(Q, q).nnz_L


def test_pin_dense_columns_pins_the_border_last():
    Qgrid, Q = backend_ordering_matrix()
    N = Q.shape[0]
    w = PinDenseColumns(mmd)
    p = ordering_permutation(Q, w)
    assert isperm(p, N)
    assert p[-1] == N - 1                          # the dense border column (test_backend_ordering.jl:47)
    be = gmrfx.MI355XBackend(Q, ordering=w, symbolic_only=True)
    assert same_elimination(be, Q, p) and be.ordering_permutation()[-1] == N - 1     # the border stays the last pivot
    # no dense columns -> transparent passthrough to the inner algorithm (:51-53)
    p2 = ordering_permutation(Qgrid, w)
    assert isperm(p2, N - 1) and np.array_equal(p2, mmd(Qgrid))
    # inner = None: libgmrfx's own nested dissection orders the sparse block, the border still goes last
    p3 = ordering_permutation(Q, PinDenseColumns())
    assert isperm(p3, N) and p3[-1] == N - 1
    # frac is honoured: with frac = 1.0 nothing is dense
    assert ordering_permutation(Q, PinDenseColumns(None, frac=1.0)) is None


def test_algorithm_object_and_vector_forms():
    _, Q = backend_ordering_matrix()
    N = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, ordering=mmd, symbolic_only=True)          # "CliqueTrees elimination algorithm"
    assert same_elimination(be, Q, mmd(Q))
    be = gmrfx.MI355XBackend(Q, ordering=list(range(N - 1, -1, -1)), symbolic_only=True)
    assert same_elimination(be, Q, np.arange(N - 1, -1, -1))
    be = gmrfx.MI355XBackend(Q, ordering="natural", symbolic_only=True)
    assert same_elimination(be, Q, np.arange(N))


def test_unknown_ordering_forms_are_rejected_not_dropped():
    _, Q = backend_ordering_matrix()
    with pytest.raises(TypeError):
        gmrfx.MI355XBackend(Q, ordering=object(), symbolic_only=True)
    with pytest.raises(TypeError):
        gmrfx.MI355XBackend(Q, ordering=3.5, symbolic_only=True)
    with pytest.raises(ValueError):
        gmrfx.MI355XBackend(Q, ordering="amd", symbolic_only=True)
    with pytest.raises(ValueError):
        gmrfx.MI355XBackend(Q, ordering=np.zeros(Q.shape[0], dtype=np.int64), symbolic_only=True)    # not a permutation
    with pytest.raises(ValueError):
        gmrfx.MI355XBackend(Q, ordering=lambda P: np.arange(3), symbolic_only=True)                   # algorithm returns junk
