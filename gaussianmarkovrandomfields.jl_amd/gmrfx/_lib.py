"""ctypes binding of libgmrfx.so -- mirrors 1:1 the `ccall`s of the Julia shim
(julia/GMRFX.jl). No compute happens in Python."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("GMRFX_LIB", os.path.join(os.path.dirname(_HERE), "libgmrfx.so"))

GMRFX_OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_FACTORIZED, ERR_NOT_POSDEF, ERR_ALLOC = 1, 2, 3, 4, 5, 6


class GmrfxOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("uplo", C.c_int32), ("ordering", C.c_int32), ("device", C.c_int32),
        ("symbolic_only", C.c_int32), ("check_posdef", C.c_int32), ("nd_leaf", C.c_int32),
        ("relax_cols", C.c_int32), ("relax_zeros", C.c_double), ("coord_dim", C.c_int32),
        ("reserved0", C.c_int32), ("coords", C.c_void_p), ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
        ("shard_min_top", C.c_int32), ("reserved1", C.c_int32),
    ]


class GmrfxStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in (
        "n", "nnz_q_tri", "nnz_l", "nnz_l_stored", "nsuper", "nlevels", "max_cols", "max_rows", "sum_rows",
        "n_small_fronts", "n_big_fronts")] + [(k, C.c_double) for k in (
        "factor_flops", "bytes_factor", "bytes_cb_arena", "bytes_device_total", "ms_symbolic", "ms_factor",
        "ms_solve", "ms_solve_fwd", "ms_solve_bwd", "ms_solve_perm", "ms_backward_solve", "ms_logdet",
        "ms_selinv")] + [("last_nrhs", C.c_int64), ("fail_col", C.c_int64), ("ms_syrk", C.c_double),
                         ("syrk_flops", C.c_double), ("syrk_launches", C.c_int64),
                         ("ms_quadform", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class GmrfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gmrfx error {code}: {msg}")
        self.code = code


class NoDeviceError(GmrfxError):
    pass


class PosDefException(GmrfxError):
    pass


_lib = None
EXPORTS = [
    "gmrfx_last_create_error", "gmrfx_last_error", "gmrfx_create", "gmrfx_destroy", "gmrfx_clone",
    "gmrfx_dense_apply_dev", "gmrfx_transpose_dev",
    "gmrfx_refactorize", "gmrfx_refactorize_dev", "gmrfx_refactorize_solve", "gmrfx_refactorize_solve_dev",
    "gmrfx_refactorize_update_solve", "gmrfx_refactorize_update_solve_dev", "gmrfx_refactorize_logpdf_dev", "gmrfx_solve", "gmrfx_solve_dev", "gmrfx_backward_solve",
    "gmrfx_backward_solve_dev", "gmrfx_logdet", "gmrfx_selinv_compute", "gmrfx_selinv_diag", "gmrfx_selinv_nnz",
    "gmrfx_selinv_csc", "gmrfx_selinv_extract", "gmrfx_get_perm", "gmrfx_get_stats", "gmrfx_symbolic_sizes",
    "gmrfx_symbolic_get", "gmrfx_get_factor_values", "gmrfx_refactorize_phase", "gmrfx_shard_info",
    "gmrfx_shard_edges", "gmrfx_shard_owner", "gmrfx_device_ptr", "gmrfx_set_stream", "gmrfx_level_times", "gmrfx_shard_dist_fronts", "gmrfx_shard_transfers",
    "gmrfx_dist_front_phase", "gmrfx_logdet_partial", "gmrfx_solve_phase",
    "gmrfx_shard_rows", "gmrfx_set_prior", "gmrfx_refactorize_update", "gmrfx_refactorize_update_dev",
    "gmrfx_quadform", "gmrfx_quadform_dev", "gmrfx_selinv_dot", "gmrfx_selinv_row_diag", "gmrfx_kl_cholesky",
    "gmrfx_selinv_row_diag_plan", "gmrfx_selinv_row_diag_apply", "gmrfx_selinv_row_diag_free",
    "gmrfx_symbolic_sweep_tasks", "gmrfx_symbolic_sweep_chunks", "gmrfx_selinv_phase", "gmrfx_host_io_plan", "gmrfx_dist_front_block",
]


def source_tree_hash() -> str:
    """sha256 (first 16 hex digits) over the library's sources -- csrc/*.{hip,cpp,h} and include/gmrfx.h, names and
    contents, in name order. The counter passes under profiles/ store it (tools/pmc_traffic.py, tools/cfg3_profile.py) and
    bench.py only quotes their traffic figures for the tree it is timing."""
    import glob, hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "..", "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "..", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(_HERE, "..", "csrc", "*.h")), key=os.path.basename)
    files.append(os.path.join(_HERE, "..", "..", "include", "gmrfx.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def lib():
    """Load libgmrfx.so; fails loudly if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise ImportError(f"{_LIBPATH} not found: build it with `make -C {os.path.dirname(_HERE)}` "
                              "(or __graft_entry__.build()); there is no non-HIP fallback")
        # One HIP runtime per process: torch ships its own libamdhip64; if libgmrfx.so pulls in the system one FIRST, a later
        # `import torch` (the *_dev entry points take torch tensors' pointers) finds no GPU. Loading torch first makes
        # libgmrfx.so resolve against the runtime already in the process. GMRFX_NO_TORCH=1 skips this (hosts without torch).
        if not os.environ.get("GMRFX_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(_LIBPATH)
        vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
        L.gmrfx_last_create_error.restype = C.c_char_p
        L.gmrfx_last_error.restype = C.c_char_p
        L.gmrfx_last_error.argtypes = [vp]
        L.gmrfx_create.argtypes = [i64, vp, vp, i32, vp, C.POINTER(GmrfxOpts), C.POINTER(vp)]
        L.gmrfx_destroy.argtypes = [vp]
        L.gmrfx_destroy.restype = None
        L.gmrfx_clone.argtypes = [vp, C.POINTER(vp)]
        L.gmrfx_refactorize.argtypes = [vp, vp, C.POINTER(i64)]
        L.gmrfx_refactorize_dev.argtypes = [vp, vp, C.POINTER(i64)]
        L.gmrfx_dense_apply_dev.argtypes = [vp, i64, i64, vp, vp, vp]
        L.gmrfx_transpose_dev.argtypes = [vp, i64, i64, vp, vp]
        L.gmrfx_refactorize_solve.argtypes = [vp, vp, vp, i64, i64, vp, i64, C.POINTER(i64)]
        L.gmrfx_refactorize_solve_dev.argtypes = [vp, vp, vp, i64, i64, vp, i64, C.POINTER(i64)]
        L.gmrfx_refactorize_logpdf_dev.argtypes = [vp, vp, vp, i64, i64, vp, vp, C.POINTER(dbl), C.POINTER(i64)]
        L.gmrfx_refactorize_update_solve.argtypes = [vp, vp, vp, i64, i64, vp, i64, C.POINTER(i64)]
        L.gmrfx_refactorize_update_solve_dev.argtypes = [vp, vp, vp, i64, i64, vp, i64, C.POINTER(i64)]
        for nm in ("gmrfx_solve", "gmrfx_solve_dev", "gmrfx_backward_solve", "gmrfx_backward_solve_dev"):
            getattr(L, nm).argtypes = [vp, vp, i64, i64, vp, i64]
        L.gmrfx_logdet.argtypes = [vp, C.POINTER(dbl)]
        L.gmrfx_selinv_compute.argtypes = [vp]
        L.gmrfx_selinv_diag.argtypes = [vp, vp]
        L.gmrfx_selinv_nnz.argtypes = [vp, C.POINTER(i64)]
        L.gmrfx_selinv_csc.argtypes = [vp, i32, vp, vp, vp]
        L.gmrfx_selinv_extract.argtypes = [vp, i64, vp, vp, i32, vp]
        L.gmrfx_get_perm.argtypes = [vp, i32, vp]
        L.gmrfx_get_stats.argtypes = [vp, C.POINTER(GmrfxStats), i32]
        L.gmrfx_symbolic_sizes.argtypes = [vp, vp]
        L.gmrfx_symbolic_get.argtypes = [vp] + [vp] * 10
        L.gmrfx_symbolic_sweep_tasks.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), vp, vp, vp]
        L.gmrfx_symbolic_sweep_chunks.argtypes = [vp, vp, C.POINTER(i64), vp, vp, vp, vp, vp]
        L.gmrfx_get_factor_values.argtypes = [vp, vp]
        L.gmrfx_refactorize_phase.argtypes = [vp, vp, i32]
        L.gmrfx_shard_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
        L.gmrfx_shard_edges.argtypes = [vp] + [vp] * 10
        L.gmrfx_selinv_phase.argtypes = [vp, i32, i32, i32]
        L.gmrfx_shard_owner.argtypes = [vp, vp, vp]
        L.gmrfx_device_ptr.argtypes = [vp, i32]
        L.gmrfx_device_ptr.restype = C.c_void_p
        L.gmrfx_set_stream.argtypes = [vp, vp, i32, i32]
        L.gmrfx_level_times.argtypes = [vp, i32, vp, i64, C.POINTER(i64)]
        L.gmrfx_shard_dist_fronts.argtypes = [vp] + [vp] * 9
        L.gmrfx_shard_transfers.argtypes = [vp] + [vp] * 7
        L.gmrfx_dist_front_phase.argtypes = [vp, vp, i32, i32, i32]
        L.gmrfx_logdet_partial.argtypes = [vp, C.POINTER(dbl)]
        L.gmrfx_solve_phase.argtypes = [vp, vp, i64, i64, vp, i64, i32]
        L.gmrfx_shard_rows.argtypes = [vp, i32, C.POINTER(i64), vp, vp, vp, vp]
        L.gmrfx_set_prior.argtypes = [vp, vp, vp, i64, i32]
        L.gmrfx_refactorize_update.argtypes = [vp, vp, C.POINTER(i64)]
        L.gmrfx_refactorize_update_dev.argtypes = [vp, vp, C.POINTER(i64)]
        L.gmrfx_quadform.argtypes = [vp, vp, vp, i64, i64, vp, vp]
        L.gmrfx_quadform_dev.argtypes = [vp, vp, vp, i64, i64, vp, vp]
        L.gmrfx_selinv_dot.argtypes = [vp, i64, vp, vp, vp, i32, C.POINTER(dbl)]
        L.gmrfx_selinv_row_diag.argtypes = [vp, i64, vp, vp, vp, i32, vp]
        L.gmrfx_kl_cholesky.argtypes = [i64, vp, i64, i32, vp, i64, vp, vp, vp, vp, i32, dbl, i32, vp, C.POINTER(i64)]
        L.gmrfx_selinv_row_diag_plan.argtypes = [vp, i64, vp, vp, i32, C.POINTER(i64)]
        L.gmrfx_selinv_row_diag_apply.argtypes = [vp, i64, vp, vp]
        L.gmrfx_selinv_row_diag_free.argtypes = [vp, i64]
        L.gmrfx_host_io_plan.argtypes = [i64, i64, i32, vp]
        L.gmrfx_dist_front_block.argtypes = [vp, i32, i32, C.POINTER(i64), C.POINTER(i64)]
        for nm in EXPORTS[2:]:
            if nm not in ("gmrfx_destroy", "gmrfx_device_ptr"):
                getattr(L, nm).restype = i32
        _lib = L
    return _lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def check(code, handle=None):
    if code == GMRFX_OK:
        return
    L = lib()
    msg = (L.gmrfx_last_error(handle) if handle else L.gmrfx_last_create_error()) or b""
    msg = msg.decode("utf-8", "replace")
    if code == ERR_INVALID_ARG:
        raise ValueError(f"gmrfx: {msg}")      # Julia: ArgumentError / DimensionMismatch
    if code == ERR_NO_DEVICE:
        raise NoDeviceError(code, msg)
    if code == ERR_NOT_POSDEF:
        raise PosDefException(code, msg)
    raise GmrfxError(code, msg)
