#!/usr/bin/env python3
"""Where a chunk's time goes inside the sweep-task kernels (csrc/sweep_chunk.hip: k_fwd_chunks / k_bwd_chunks), from s_memtime
stamps compiled in with -DGMRFX_CYC (`make -C gaussianmarkovrandomfields.jl_amd cyc` -> libgmrfx_cyc.so; the product library has
none of it). One 64-RHS solve at cfg 2 (AB_NRHS / SWL_GRID in the environment change that); per kernel and row-tile slot:
cycles summed over all waves by category, as a share of the slot's wave cycles and per chunk record.

    python3 tools/chunk_cycles.py > gpurun_out/chunk_cycles.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd")
os.environ["GMRFX_LIB"] = os.path.join(PKG, "libgmrfx_cyc.so")
sys.path.insert(0, PKG)
import numpy as np, torch        # noqa: E402
import gmrfx                      # noqa: E402
from gmrfx import spde, _lib      # noqa: E402

CATS = ["prologue", "issue next", "wait operands", "y / k-tiles", "apply / x", "barrier", "epilogue"]


def main():
    grid = int(os.environ.get("SWL_GRID", "1000"))
    nr = int(os.environ.get("AB_NRHS", "64"))
    mesh = spde.grid_mesh_2d(grid, grid, jitter=0.25, seed=0)
    Q = spde.matern_precision(mesh, 0, 0.2)
    n = Q.shape[0]
    dev = torch.device("cuda", 0)
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
    L = _lib.lib()
    L.gmrfx_debug_chunk_cycles.argtypes = [C.c_void_p, C.c_int]
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
    d_B = torch.randn((nr, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
    d_X = torch.empty_like(d_B)
    torch.cuda.synchronize()
    be.refactorize_dev(d_nz.data_ptr())
    for _ in range(3):
        be.solve_dev(d_B.data_ptr(), n, nr, d_X.data_ptr(), n)
    torch.cuda.synchronize()
    assert L.gmrfx_debug_chunk_cycles(None, 1) == 0
    be.solve_dev(d_B.data_ptr(), n, nr, d_X.data_ptr(), n)
    torch.cuda.synchronize()
    st = be.stats()
    out = np.zeros((2, 4, 10), np.uint64)
    assert L.gmrfx_debug_chunk_cycles(out.ctypes.data_as(C.c_void_p), 0) == 0
    cap, tf, tl, _ = be.sweep_tasks()
    print(f"# tools/chunk_cycles.py: cfg 2 grid {grid}, {nr} right-hand sides, {len(tf)} sweep tasks x {(nr + 15) // 16} column slices = "
          f"{len(tf) * ((nr + 15) // 16)} workgroups of 4 waves (slot w = row-tile slot); INSTRUMENTED build (stamps + explicit vmcnt waits):")
    print(f"# forward sweep {st['ms_solve_fwd']:.3f} ms, backward {st['ms_solve_bwd']:.3f} ms with it")
    print("# cycles are s_memtime ticks summed over the waves of a slot; share = of the slot's summed wave cycles; /rec = per chunk record passed")
    for k, name in enumerate(("k_fwd_chunks<16,1>", "k_bwd_chunks<16,1,3>")):
        print(f"\n== {name}")
        print(f"{'slot':>4s} {'records':>10s} {'with work':>10s} " + " ".join(f"{c:>16s}" for c in CATS) + f" {'total/rec':>10s}")
        tot_all = np.zeros(7)
        for w in range(4):
            a = out[k, w].astype(np.float64)
            tot = a[:7].sum()
            rec = max(a[7], 1.0)
            tot_all += a[:7]
            print(f"{w:4d} {int(a[7]):10d} {int(a[8]):10d} " + " ".join(f"{100 * a[c] / tot:7.1f}% {a[c] / rec:7.0f}" for c in range(7)) + f" {tot / rec:10.0f}")
        t = tot_all.sum()
        print(f"{'all':>4s} {'':10s} {'':10s} " + " ".join(f"{100 * tot_all[c] / t:7.1f}% {'':7s}" for c in range(7)))
    be.close()


if __name__ == "__main__":
    main()
