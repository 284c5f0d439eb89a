"""ctypes binding of oracle/supernodal_cpu.c -- the multi-threaded supernodal CPU factor + multi-RHS solve that
bench.py times as `cpu_baseline` (kind "port"). TEST / BENCH INFRASTRUCTURE ONLY.

Dense kernels = the OpenBLAS inside the scipy wheel (LP64 `scipy_cblas_*` / `scipy_LAPACKE_dpotrf`), looked up at run
time and handed to the C code as function pointers. The symbolic structure (supernodes, row lists, relative indices,
panel layout, levels, Q scatter map) is the one libgmrfx's host analysis exports (gmrfx_symbolic_get), so CPU and GPU
factor the same P Q P' with the same supernode partition."""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_BLAS = None


def _blas():
    global _BLAS
    if _BLAS is None:
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas-*.so"))
        if not cands:
            raise RuntimeError("no LP64 OpenBLAS found inside the scipy wheel")
        _BLAS = C.CDLL(cands[0])
    return _BLAS


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsncpu.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libsncpu.so"])
        L = C.CDLL(path)
        L.sncpu_create.restype = C.c_void_p
        L.sncpu_create.argtypes = [C.c_int, C.c_int64] + [C.c_void_p] * 8 + [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
        L.sncpu_free.argtypes = [C.c_void_p]
        L.sncpu_factor.restype = C.c_int64
        L.sncpu_factor.argtypes = [C.c_void_p]
        L.sncpu_logdet.restype = C.c_double
        L.sncpu_logdet.argtypes = [C.c_void_p]
        L.sncpu_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _LIB = L
    return _LIB


class SupernodalCPU:
    """sy: gmrfx.SymbolicInfo of a (symbolic-only) MI355XBackend, perm: its elimination order."""

    def __init__(self, sy, perm, n: int, nthreads: int | None = None):
        self.n, self.perm = n, np.asarray(perm, dtype=np.int64)
        self.nthreads = int(nthreads or min(os.cpu_count() or 1, 16))
        B = _blas()
        fp = lambda name: C.cast(getattr(B, name), C.c_void_p)
        self._keep = [np.ascontiguousarray(a, dtype=np.int64) for a in
                      (sy.super_first, sy.row_ptr, sy.rows, sy.rel, sy.super_parent, sy.panel_ptr, sy.panel_ld, sy.level)]
        self.q_src = np.asarray(sy.q_src, dtype=np.int64)
        self.q_dst = np.asarray(sy.q_dst, dtype=np.int64)
        self.L = np.zeros(int(sy.panel_ptr[-1]), dtype=np.float64)
        ns = len(sy.super_parent)
        self._h = lib().sncpu_create(ns, n, *[a.ctypes.data for a in self._keep], self.L.ctypes.data,
                                     fp("scipy_cblas_dgemm"), fp("scipy_cblas_dsyrk"), fp("scipy_cblas_dtrsm"),
                                     fp("scipy_LAPACKE_dpotrf"), fp("scipy_openblas_set_num_threads"), self.nthreads)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sncpu_free(self._h)
            self._h = None

    def factorize(self, nzval) -> int:
        """Numeric factorisation on the fixed pattern (refactorize!): returns -1 or the failing column."""
        self.L[:] = 0.0
        self.L[self.q_dst] = np.asarray(nzval, dtype=np.float64)[self.q_src]
        return int(lib().sncpu_factor(self._h))

    def logdet(self) -> float:
        return float(lib().sncpu_logdet(self._h))

    def solve(self, B, mode: int = 0):
        """Q X = B (mode 0) or X = P' L^-T B (mode 1); B: n or n x k, original ordering."""
        B = np.asarray(B, dtype=np.float64)
        Bm = B.reshape(self.n, -1)
        X = np.ascontiguousarray(Bm[self.perm] if mode == 0 else Bm)        # row-major n x k, elimination order
        lib().sncpu_solve(self._h, X.ctypes.data, X.shape[1], mode)
        out = np.empty_like(X)
        out[self.perm] = X
        return out.reshape(B.shape)
