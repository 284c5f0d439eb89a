#!/usr/bin/env python3
"""Knock-out timing of the sweep-task kernels (csrc/sweep_chunk.hip built with -DGMRFX_VAR: `make -C gaussianmarkovrandomfields.jl_amd var`
-> libgmrfx_var.so): the forward / backward sweep of a 64-RHS solve at cfg 2 with parts of the two task kernels switched off by a
run-time flag word (results are WRONG in every variant but the first; only the times mean anything), and the histogram of the SIMD
each row-tile slot's wave runs on. The task kernels are the first launch of the forward and the last of the backward sweep; the
level kernels behind them are the same in every variant, so differences between variants are the task kernels'.

    python3 tools/chunk_variants.py > gpurun_out/chunk_variants.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd")
os.environ["GMRFX_LIB"] = os.path.join(PKG, "libgmrfx_var.so")
sys.path.insert(0, PKG)
import numpy as np, torch        # noqa: E402
import gmrfx                      # noqa: E402
from gmrfx import spde, _lib      # noqa: E402

VARIANTS = [(0, "production code path (flags 0)"),
            (1, "no operand requests beyond a task's first chunk (LDS / MFMA / barrier chain only)"),
            (2, "no arithmetic (operand requests + barriers only)"),
            (4, "no barriers"),
            (1 | 4, "no requests, no barriers (arithmetic chain of each wave alone)"),
            (2 | 4, "no arithmetic, no barriers (requests only)"),
            (1 | 2 | 4, "prologue + epilogue only (record walk, X in / out)")]


def main():
    grid = int(os.environ.get("SWL_GRID", "1000"))
    nr = int(os.environ.get("AB_NRHS", "64"))
    mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
    Q = spde.matern_precision(mesh, 0, 0.2)
    n = Q.shape[0]
    dev = torch.device("cuda", 0)
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
    L = _lib.lib()
    L.gmrfx_debug_chunk_variant.argtypes = [C.c_int, C.c_void_p]
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
    d_B = torch.randn((nr, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
    d_X = torch.empty_like(d_B)
    torch.cuda.synchronize()
    be.refactorize_dev(d_nz.data_ptr())
    print(f"# tools/chunk_variants.py: cfg 2 grid {grid}, {nr} right-hand sides; forward / backward sweep ms (median of 8 solves) per variant of the task kernels;")
    print("# the task kernels' share is the difference to the last line (prologue + epilogue only) plus that line's own task time")
    base = None
    for flags, what in VARIANTS:
        assert L.gmrfx_debug_chunk_variant(flags, None) == 0
        f, b = [], []
        for k in range(10):
            be.solve_dev(d_B.data_ptr(), n, nr, d_X.data_ptr(), n)
            st = be.stats()
            if k >= 2:
                f.append(st["ms_solve_fwd"]); b.append(st["ms_solve_bwd"])
        mf, mb = float(np.median(f)), float(np.median(b))
        if base is None:
            base = (mf, mb)
        print(f"flags {flags:2d}: forward {mf:6.3f} ms ({mf - base[0]:+.3f})  backward {mb:6.3f} ms ({mb - base[1]:+.3f})   {what}")
    # SIMD histogram of the production path
    hist = np.zeros((2, 4, 4), np.uint64)
    assert L.gmrfx_debug_chunk_variant(16, hist.ctypes.data_as(C.c_void_p)) == 0       # (clears)
    be.solve_dev(d_B.data_ptr(), n, nr, d_X.data_ptr(), n)
    torch.cuda.synchronize()
    assert L.gmrfx_debug_chunk_variant(0, hist.ctypes.data_as(C.c_void_p)) == 0
    print("\n# SIMD (HW_ID bits 5:4) of the wave of every row-tile slot, counted over the workgroups of one launch")
    for k, name in enumerate(("k_fwd_chunks", "k_bwd_chunks")):
        for w in range(4):
            print(f"{name} slot {w}: " + "  ".join(f"SIMD{s_} {int(hist[k, w, s_]):7d}" for s_ in range(4)))
    be.close()


if __name__ == "__main__":
    main()
