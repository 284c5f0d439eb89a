#!/usr/bin/env python3
"""What in the forward TASK kernel costs the factorisation beside it in the pipelined call -- its memory traffic or its occupancy
(LDS / wave slots)? The pipelined step (gmrfx_refactorize_solve_dev, cfg 2, 64 right-hand sides) on the knock-out build of the
chunk kernels (`make var`: libgmrfx_var.so; tools/chunk_variants.py): the task kernel with its operand requests, its arithmetic or
both switched off (WRONG results: timing only), against the factorisation alone. Device times per step from the library's events:
ms_factor = the factorisation inside the call, ms_solve = what is left behind it.

    python3 tools/pipeline_variants.py > gpurun_out/r06_pipeline_variants.txt"""
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

mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, 0, 0.2)
n, nrhs = Q.shape[0], 64
be = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
L = _lib.lib()
L.gmrfx_debug_chunk_variant.argtypes = [C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
d_B = torch.randn((nrhs, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
d_X = torch.empty_like(d_B)
torch.cuda.synchronize()


def med(fn, k=14):
    v = []
    for i in range(k):
        fn()
        s = be.stats()
        if i >= 3:
            v.append((s["ms_factor"], s["ms_solve"]))
    return np.median(np.asarray(v), axis=0)


alone = med(lambda: be.refactorize_dev(d_nz.data_ptr()))[0]
print(f"# tools/pipeline_variants.py, cfg 2, 64 right-hand sides; factorisation ALONE (gmrfx_refactorize_dev): {alone:.3f} ms")
print("# pipelined call with variants of the forward task kernel k_fwd_chunks (the one launch of the forward sweep that runs beside the top of the factorisation's middle):")
for flags, what in ((0, "production"), (1, "no operand requests (LDS / MFMA chain only: occupancy without traffic)"),
                    (2, "no arithmetic (requests + barriers: traffic without the chain)"), (7, "prologue + epilogue only (X in, W out)")):
    assert L.gmrfx_debug_chunk_variant(flags, None) == 0
    f, s = med(lambda: be.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, nrhs, d_X.data_ptr(), n))
    print(f"flags {flags}: factorisation inside the call {f:7.3f} ms (+{f - alone:.3f} over alone), behind it {s:6.3f} ms, step {f + s:7.3f} ms   {what}")
assert L.gmrfx_debug_chunk_variant(0, None) == 0
be.close()
