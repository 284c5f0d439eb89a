"""CPU: libgmrfx.so loads and exports every symbol include/gmrfx.h declares; struct layouts of
the ctypes mirror match sizeof on the C side is implied by struct_size round-trips."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import gmrfx
from gmrfx import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gmrfx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gmrfx_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/gmrfx.h but not exported"
    assert sorted(_lib.EXPORTS) == syms


def test_stats_struct_roundtrip_and_symbolic_only_handle():
    import scipy.sparse as sp
    Q = sp.csc_matrix(np.array([[4.0, 1, 0], [1, 3, 1], [0, 1, 5]]))
    opts = _lib.GmrfxOpts()
    opts.struct_size = C.sizeof(_lib.GmrfxOpts)
    opts.symbolic_only = 1
    opts.device = -1
    h = C.c_void_p()
    colptr = np.ascontiguousarray(Q.indptr, dtype=np.int64) + 1       # 1-based, as Julia passes it
    rowval = np.ascontiguousarray(Q.indices, dtype=np.int64) + 1
    rc = _lib.lib().gmrfx_create(3, _lib.ptr(colptr), _lib.ptr(rowval), 1, None, C.byref(opts), C.byref(h))
    assert rc == 0, _lib.lib().gmrfx_last_create_error()
    st = _lib.GmrfxStats()
    assert _lib.lib().gmrfx_get_stats(h, C.byref(st), C.sizeof(_lib.GmrfxStats)) == 0
    assert st.n == 3 and st.nnz_l >= 5 and st.fail_col == -1
    perm = np.zeros(3, np.int64)
    assert _lib.lib().gmrfx_get_perm(h, 1, _lib.ptr(perm)) == 0
    assert sorted(perm.tolist()) == [1, 2, 3]
    # numeric entry points refuse to run without a device: no CPU fallback
    out = C.c_double()
    assert _lib.lib().gmrfx_logdet(h, C.byref(out)) == _lib.ERR_NO_DEVICE
    assert b"GPU-only" in _lib.lib().gmrfx_last_error(h)
    _lib.lib().gmrfx_destroy(h)


def test_bad_arguments_are_reported_not_crashed():
    h = C.c_void_p()
    colptr = np.array([0, 1, 5], dtype=np.int64)     # rowval out of range
    rowval = np.array([0, 7, 1, 1, 1], dtype=np.int64)
    rc = _lib.lib().gmrfx_create(2, _lib.ptr(colptr), _lib.ptr(rowval), 0, None, None, C.byref(h))
    assert rc == _lib.ERR_INVALID_ARG and h.value is None
    assert b"range" in _lib.lib().gmrfx_last_create_error()


def test_kl_cholesky_validates_and_fails_loudly_without_a_gpu():
    # no CPU fallback anywhere: argument errors are reported before the device is touched, and on a box without
    # a GPU the numeric entry point returns a HIP error instead of computing on the host
    import scipy.sparse as sp
    from gmrfx import klchol
    with pytest.raises(ValueError):
        klchol._run(np.eye(3), [0, 1, 2, 3], [0, 1], [5], [0, 1], [0], 1e-6, -1)          # row index out of range
    with pytest.raises(ValueError):
        klchol._run(np.eye(3), [0, 2, 3, 4], [0, 1], [0], [0, 1], [0], 1e-6, -1)          # column longer than its task
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(gmrfx.GmrfxError):
            klchol.sparse_approximate_cholesky_inplace(np.eye(3), sp.csc_matrix(np.tril(np.ones((3, 3)))))


@pytest.mark.parametrize("download", [0, 1])
def test_host_io_slices_fit_their_ring_slot_and_tile_the_array(download):
    # the staging ring of the host entry points (device.cpp host_upload / host_download): every slice must fit its slot of the
    # page-locked buffer whatever n is -- round 5 sized the slot by n once a column exceeded a slice (n > 2^21: a 1500 x 1500 grid,
    # a 130^3 mesh) but clamped the ring to 128 MB, so slots 6 / 7 lay behind the allocation. Arithmetic only: no device needed.
    L = _lib.lib()
    ring = 8
    for n, nrhs in [(1, 1), (9, 3), (10 ** 6, 64), (2 ** 21, 64), (2 ** 21 + 1, 64), (2_250_000, 8), (2_197_000, 70), (10 ** 7, 5),
                    (3 * 2 ** 21 + 5, 9), (2 ** 20 + 3, 1), (123_457, 300)]:
        plan = np.zeros(6, np.int64)
        assert L.gmrfx_host_io_plan(n, nrhs, download, _lib.ptr(plan)) == 0
        cols_per, ppc, rows_per, slot, nsl, reserve = plan.tolist()
        slice_doubles = (16 << 20) // 8 // (2 if download else 1)
        assert slot <= max(slice_doubles, 1) and reserve == min(nsl, ring) * slot and reserve <= ring * slice_doubles
        covered = 0
        for k in range(nsl):
            if ppc == 1:
                j0 = k * cols_per
                nc, r0, nr = min(cols_per, nrhs - j0), 0, n
            else:
                j0, nc = k // ppc, 1
                r0 = (k % ppc) * rows_per
                nr = max(0, min(rows_per, n - r0))
            assert 0 <= j0 and j0 + nc <= nrhs and 0 <= r0 and r0 + nr <= n and nc >= 1
            assert nc * nr <= slot and ((k % ring) + 1) * slot <= reserve
            covered += nc * nr
        assert covered == n * nrhs
