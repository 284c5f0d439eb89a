"""CPU, gloo, world_size 2 and 3: the SHARDED-factorisation path (gmrfx/shard.py). Every rank
runs the same symbolic analysis with its own shard_rank, factors the fronts it owns with the test's host
walk (tests/mf_hostsim.py -- this container has no GPU), the contribution blocks of the subtree roots
travel to rank 0 over the process group, rank 0 factors the top fronts, and log det Q is an all-reduce of
the partial sums. Checks the plan invariants and the result against the unsharded walk."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gmrfx
    from gmrfx import spde
    from mf_hostsim import HostSim
    mesh = spde.grid_mesh_2d(40, 40, jitter=0.2)
    Q = spde.matern_precision(mesh, 0, 0.4)
    n = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True, shard_rank=rank, shard_world=world)
    sy = be.symbolic()
    owner = be.shard_owner()
    par = sy.super_parent
    ns = len(par)
    info = be.shard_info()
    # ---- plan invariants -------------------------------------------------------------------------
    assert ((owner >= -1) & (owner < world)).all() and (owner == -1).sum() == info["n_top_fronts"]
    for s in range(ns):
        p = par[s]
        if p >= 0:
            assert sy.level[p] > sy.level[s]
            assert owner[p] == -1 or owner[p] == owner[s]        # subtrees are closed downwards
            if owner[s] == -1:
                assert owner[p] == -1                            # the top is closed upwards
        if owner[s] == -1:
            assert sy.level[s] >= info["shard_level"]
        else:
            assert sy.level[s] < info["shard_level"]
    roots = [s for s in range(ns) if owner[s] >= 0 and par[s] >= 0 and owner[par[s]] == -1]
    assert len(roots) == info["n_cb_blocks"]
    # ---- sharded host walk -----------------------------------------------------------------------
    sim = HostSim(sy, n, np.asarray(Q.data))
    mine = lambda s: owner[s] == rank or (owner[s] == -1 and rank == 0)
    cb = {}

    def factor_front(s):
        c, r = sim.c[s], sim.r[s]
        F = np.zeros((r, r))
        P = sim.panel(sim.L, s)
        F[:, :c] = P
        for d in sim.children[s]:
            rel = sim.rel(d)
            F[np.ix_(rel, rel)] += cb.pop(d)
        F = np.tril(F); F = F + np.tril(F, -1).T
        L11 = np.linalg.cholesky(F[:c, :c])
        L21 = np.linalg.solve(L11, F[c:, :c].T).T
        P[:c, :] = np.tril(L11); P[c:, :] = L21
        cb[s] = F[c:, c:] - L21 @ L21.T

    for s in sim.order:                                   # phase 0: own subtrees (levels < shard_level)
        if owner[s] >= 0 and mine(s):
            factor_front(s)
    for d in roots:                                       # exchange: subtree-root CBs -> rank 0
        m = int(sim.r[d] - sim.c[d])
        if owner[d] == 0:
            continue
        if rank == owner[d]:
            dist.send(torch.from_numpy(np.ascontiguousarray(cb.pop(d))), dst=0, tag=int(d))
        elif rank == 0:
            buf = torch.empty((m, m), dtype=torch.float64)
            dist.recv(buf, src=int(owner[d]), tag=int(d))
            cb[d] = buf.numpy()
    for s in sim.order:                                   # phase 1: the top (rank 0)
        if owner[s] == -1 and mine(s):
            factor_front(s)
    # ---- sharded solve: own forward, W of the subtree roots -> rank 0, the top on rank 0, x of the top
    #      fronts -> everybody, own backward, owned x -> rank 0 -----------------------------------------
    k = 3
    Bp = np.random.default_rng(5).standard_normal((n, k))          # right-hand sides in elimination order
    X = Bp.copy()
    W = {}

    def fwd_front(s):
        c, r = sim.c[s], sim.r[s]
        f = np.zeros((r, k)); rows = sim.rows(s)
        f[:c] = X[rows[:c]]
        for d in sim.children[s]:
            f[sim.rel(d)] += W.pop(d)
        P = sim.panel(sim.L, s)
        y = np.linalg.solve(np.tril(P[:c]), f[:c])
        X[rows[:c]] = y
        W[s] = f[c:] - P[c:] @ y

    def bwd_front(s):
        c = sim.c[s]; rows = sim.rows(s); P = sim.panel(sim.L, s)
        X[rows[:c]] = np.linalg.solve(np.tril(P[:c]).T, X[rows[:c]] - P[c:].T @ X[rows[c:]])

    for s in sim.order:
        if owner[s] >= 0 and mine(s):
            fwd_front(s)
    for d in roots:
        m = int(sim.r[d] - sim.c[d])
        if owner[d] == 0:
            continue
        if rank == owner[d]:
            dist.send(torch.from_numpy(np.ascontiguousarray(W.pop(d))), dst=0, tag=1000 + int(d))
        elif rank == 0:
            buf = torch.empty((m, k), dtype=torch.float64)
            dist.recv(buf, src=int(owner[d]), tag=1000 + int(d))
            W[d] = buf.numpy()
    tops = [s for s in sim.order if owner[s] == -1]
    if rank == 0:
        for s in tops:
            fwd_front(s)
        for s in tops[::-1]:
            bwd_front(s)
    for s in tops:                                        # x of the top fronts' columns: rank 0 -> all
        rows = sim.rows(s)[:sim.c[s]]
        buf = torch.from_numpy(np.ascontiguousarray(X[rows]))
        dist.broadcast(buf, src=0)
        X[rows] = buf.numpy()
    for s in sim.order[::-1]:
        if owner[s] >= 0 and mine(s):
            bwd_front(s)
    for s in range(ns):                                   # owned x -> rank 0 (per front here; per subtree on the device)
        if owner[s] <= 0:
            continue
        rows = sim.rows(s)[:sim.c[s]]
        if rank == owner[s]:
            dist.send(torch.from_numpy(np.ascontiguousarray(X[rows])), dst=0, tag=2000 + s)
        elif rank == 0:
            buf = torch.empty((len(rows), k), dtype=torch.float64)
            dist.recv(buf, src=int(owner[s]), tag=2000 + s)
            X[rows] = buf.numpy()
    solve_err = None
    if rank == 0:
        perm = be.ordering_permutation()
        Qp = Q.toarray()[np.ix_(perm, perm)]
        solve_err = float(np.abs(Qp @ X - Bp).max() / np.abs(Bp).max())
    part = 2.0 * sum(np.log(np.diag(sim.panel(sim.L, s)[:sim.c[s]])).sum() for s in range(ns) if mine(s))
    t = torch.tensor([part], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if rank == 0:
        ref = HostSim(gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True).symbolic(), n, np.asarray(Q.data)).factor().logdet()
        dense = np.linalg.slogdet(Q.toarray())[1]
        q.put((float(t.item()), ref, dense, info, [int((owner == kk).sum()) for kk in range(-1, world)], solve_err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_factor_plan_and_logdet_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29610 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = q.get(timeout=300)
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    sharded, ref, dense, info, counts, solve_err = got
    assert solve_err < 1e-10
    assert abs(sharded - ref) <= 1e-10 * abs(ref)
    assert abs(sharded - dense) <= 1e-9 * abs(dense)
    assert info["n_top_fronts"] >= 1 and all(c > 0 for c in counts[1:])      # every rank owns something
