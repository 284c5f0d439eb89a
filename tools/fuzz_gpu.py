#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: many small/medium SPD patterns of different shapes against dense
LAPACK identities (solve, logdet, selinv diag, backward-solve covariance identity). usage: fuzz_gpu.py [ncases] [seed]
`fuzz_gpu.py big [ncases] [seed]`: 2-D / 3-D meshes of 2e4-2e5 nodes instead (level kernels, sweep tasks, narrow and wide passes) --
no dense reference at that size: normwise backward errors of solves with 1 / 3 / 16 / 33 / 64 / 70 right-hand sides, the backward-solve identity
s'Qs = z'z, and every pass width against the 64-column pass of the same columns (1e-10)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gmrfx
from gmrfx import spde
from mirror import GMRFWorkspace

big = len(sys.argv) > 1 and sys.argv[1] == "big"
if big: sys.argv.pop(1)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rand_case(i):
    kind = i % 6
    if kind == 0:      # random sparse SPD
        n = int(rng.integers(1, 900)); dens = float(rng.uniform(0.002, 0.08))
        return f"rand n={n} d={dens:.3f}", spde.random_spd_precision(n, dens, seed=int(rng.integers(1 << 30))), {}
    if kind == 1:      # 2-D Matern on a ragged grid with coordinates
        a, b = int(rng.integers(3, 70)), int(rng.integers(3, 70))
        m = spde.grid_mesh_2d(a, b, jitter=float(rng.uniform(0, 0.3)), seed=int(rng.integers(1 << 30)))
        return f"matern2d {a}x{b}", spde.matern_precision(m, int(rng.integers(0, 2)), float(rng.uniform(0.1, 0.6))), {"coords": m.points}
    if kind == 2:      # 3-D
        a, b, c = (int(rng.integers(2, 12)) for _ in range(3))
        m = spde.grid_mesh_3d(a, b, c)
        return f"matern3d {a}x{b}x{c}", spde.matern_precision(m, 0, float(rng.uniform(0.3, 0.8))), {"coords": m.points}
    if kind == 3:      # block diagonal (forest) of random blocks
        blocks = [spde.random_spd_precision(int(rng.integers(1, 200)), 0.1, seed=int(rng.integers(1 << 30))) for _ in range(int(rng.integers(2, 6)))]
        return "blockdiag", sp.block_diag(blocks, format="csc"), {}
    if kind == 4:      # banded / chain with natural ordering
        n = int(rng.integers(2, 1500)); bw = min(int(rng.integers(1, 12)), n - 1)
        A = sp.diags([rng.uniform(-0.2, 0.2, n - k) for k in range(1, bw + 1)], list(range(1, bw + 1)), shape=(n, n))
        A = A + A.T + sp.diags(np.full(n, 2.0 * bw))
        return f"band n={n} bw={bw}", sp.csc_matrix(A), {"ordering": "natural"} if i % 12 == 4 else {}
    n = int(rng.integers(65, 400))     # dense-ish: one big front with several 64-blocks
    G = rng.standard_normal((n, 3 * n))
    return f"dense n={n}", sp.csc_matrix(G @ G.T / (3 * n) + np.eye(n)), {}


worst = 0.0
t0 = time.time()
if big:
    for i in range(ncases):
        if i % 3 == 2:
            a = int(rng.integers(20, 44)); m = spde.grid_mesh_3d(a, a + int(rng.integers(0, 6)), a); name = f"matern3d {a}^3"
            Q = spde.matern_precision(m, 0, float(rng.uniform(0.3, 0.8)))
        else:
            a, b = int(rng.integers(140, 460)), int(rng.integers(140, 460))
            m = spde.grid_mesh_2d(a, b, jitter=float(rng.uniform(0, 0.3)), seed=int(rng.integers(1 << 30))); name = f"matern2d {a}x{b}"
            sm, rg = int(rng.integers(0, 2)), float(rng.uniform(0.1, 0.6))
            name += f" s{sm} r{rg:.2f}"
            Q = spde.matern_precision(m, sm, rg)
        Q = sp.csc_matrix(Q); n = Q.shape[0]
        be = gmrfx.MI355XBackend(Q, coords=m.points)
        B = rng.standard_normal((n, 70))
        X64 = be.backend_solve(B[:, :64])
        qn = abs(Q).sum(axis=0).max()            # ||Q||_1: residuals are judged as NORMWISE BACKWARD ERRORS (alpha = 3 precisions on
        bwerr = lambda X, Bk: np.linalg.norm(Q @ X - Bk) / (qn * np.linalg.norm(X) + np.linalg.norm(Bk))   # 400^2 nodes have cond ~ 1e9)
        errs = [bwerr(X64, B[:, :64])]
        for k in (1, 3, 16, 33, 70):
            Xk = be.backend_solve(B[:, :k])
            errs.append(bwerr(Xk, B[:, :k]))
            kk = min(k, 64)
            errs.append(np.abs(Xk[:, :kk] - X64[:, :kk]).max() / np.abs(X64).max())
        z = rng.standard_normal((n, 5))
        S5 = be.backend_backward_solve(z)
        errs.append(1e-2 * np.abs(np.einsum("ij,ij->j", S5, Q @ S5) / np.einsum("ij,ij->j", z, z) - 1.0).max())     # (the identity feels cond(Q)^(1/2): judged at 1e-8)
        s1 = be.backend_backward_solve(z[:, 0])
        errs.append(np.abs(np.ravel(s1) - S5[:, 0]).max() / np.abs(S5).max())
        err = max(errs); worst = max(worst, err)
        print(f"{i:3d} {name:32s} n={n:7d} worst {err:.2e}" + ("" if err < 1e-10 else "   <-- FAIL " + " ".join(f"{e:.1e}" for e in errs)), flush=True)
        be.close()
    print(f"worst {worst:.2e} over {ncases} big cases in {time.time() - t0:.1f}s")
    sys.exit(0 if worst < 1e-10 else 1)
for i in range(ncases):
    name, Q, kw = rand_case(i)
    Q = sp.csc_matrix(Q); n = Q.shape[0]
    ws = GMRFWorkspace(Q, **kw)
    Qd = Q.toarray()
    nrhs = int(rng.choice([1, 2, 17, 64, 65]))
    B = rng.standard_normal((n, nrhs))
    X = ws.workspace_solve(B[:, 0] if nrhs == 1 else B).reshape(n, -1)
    Xd = np.linalg.solve(Qd, B)
    e1 = np.abs(X - Xd).max() / max(np.abs(Xd).max(), 1e-300)
    e2 = abs(ws.logdet() - np.linalg.slogdet(Qd)[1]) / max(1.0, abs(np.linalg.slogdet(Qd)[1]))
    Qi = np.linalg.inv(Qd)
    e3 = np.abs(ws.selinv_diag() - np.diag(Qi)).max() / np.abs(np.diag(Qi)).max()
    z = rng.standard_normal(n)
    s = ws.backward_solve(z)                  # s = P' L^-T z  =>  Q s = P' L z ... check via ||L' P s - z|| implicitly: s' Q s = z' z
    e4 = abs(s @ (Qd @ s) - z @ z) / (z @ z)
    err = max(e1, e2, e3, e4)
    worst = max(worst, err)
    flag = "" if err < 1e-8 else "   <-- FAIL"
    print(f"{i:3d} {name:28s} n={n:5d} nrhs={nrhs:2d} solve={e1:.1e} logdet={e2:.1e} selinv={e3:.1e} bwd={e4:.1e}{flag}", flush=True)
    ws.backend.close()
print(f"worst {worst:.2e} over {ncases} cases in {time.time() - t0:.1f}s")
sys.exit(0 if worst < 1e-8 else 1)
