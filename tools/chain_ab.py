"""A/B of the persistent panel chain (csrc/panel_chain.hip; GMRFX_CHAIN_MAX_FRONTS, 0 = the launch chain of rounds 1-3): bit equality
of the factor on a few meshes and the factorisation time at cfg 2 (each setting in a child process: the switch is read once).
usage: python tools/chain_ab.py [check|time]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))


def child(mode):
    import hashlib
    import time
    import numpy as np
    import torch
    import gmrfx
    from gmrfx import spde
    if mode == "check":
        cases = [("2d_300x290", spde.grid_mesh_2d(300, 290, jitter=0.25, seed=5), 0.2), ("2d_120x500", spde.grid_mesh_2d(120, 500, jitter=0.2, seed=2), 0.3),
                 ("3d_26x25x24", spde.grid_mesh_3d(26, 25, 24), 0.4)]
        for name, mesh, rng_ in cases:
            Q = spde.matern_precision(mesh, 0, rng_)
            be = gmrfx.MI355XBackend(Q, coords=mesh.points)
            f = be.factor_values()
            B = np.random.default_rng(1).standard_normal((Q.shape[0], 3))
            X = be.backend_solve(B)
            res = np.linalg.norm(Q @ X - B) / np.linalg.norm(B)
            print(name, hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest()[:16], f"{be.compute_logdet():.15e}", f"resid {res:.2e}", "info", be.last_info, flush=True)
            be.close()
    else:
        mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
        Q = spde.matern_precision(mesh, smoothness=0, range_=0.2)
        n = Q.shape[0]
        be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
        dev = torch.device("cuda", 0)
        d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
        d_B = torch.randn((64, n), generator=torch.Generator(device="cpu").manual_seed(1), dtype=torch.float64).to(dev)
        d_X = torch.empty_like(d_B)
        tf, tp = [], []
        for _ in range(12):
            be.refactorize_dev(d_nz.data_ptr()); tf.append(be.stats()["ms_factor"])
        for _ in range(12):
            t0 = time.perf_counter()
            be.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, 64, d_X.data_ptr(), n)
            tp.append(1e3 * (time.perf_counter() - t0))
        X = d_X.cpu().numpy().T
        Bn = d_B.cpu().numpy().T
        res = np.linalg.norm(Q @ X - Bn) / np.linalg.norm(Bn)
        print(f"factor alone {np.median(tf):.3f} ms; pipelined step {np.median(tp):.3f} ms (factor part {be.stats()['ms_factor']:.3f}); logdet {be.compute_logdet():.15e}; resid {res:.2e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        child(sys.argv[1])
    else:
        mode = sys.argv[1] if len(sys.argv) > 1 else "check"
        for v in ("0", "32"):
            print(f"== GMRFX_CHAIN_MAX_FRONTS={v}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), mode, "child"], env={**os.environ, "GMRFX_CHAIN_MAX_FRONTS": v}, timeout=300)
