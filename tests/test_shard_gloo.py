"""CPU, gloo, world_size 2, 3 and 5: the SHARDED-factorisation path (gmrfx/shard.py). Every rank runs the
same symbolic analysis with its own shard_rank and factors the fronts it owns with the test's host walk
(tests/mf_hostsim.py -- this container has no GPU): its subtrees first, then the top one level per phase, each top
front on its one owner, the contribution blocks / update vectors of the cross-rank tree edges (gmrfx_shard_edges)
travelling point-to-point src -> dst before the phase that needs them, x of the top fronts broadcast by their
owners, log det Q an all-reduce of the partial sums. Checks the plan invariants (group ownership, edges = exactly
the owner-crossing tree edges, levels) and the results against the unsharded walk and dense LAPACK."""
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
    owner, is_top = be.shard_owner(with_top=True)
    par = sy.super_parent
    ns = len(par)
    info = be.shard_info()
    E = be.shard_edges()
    L0, K = info["shard_level"], info["n_top_levels"]
    # ---- plan invariants -------------------------------------------------------------------------
    assert ((owner >= 0) & (owner < world)).all() and is_top.sum() == info["n_top_fronts"]
    kids = [[] for _ in range(ns)]
    for s in range(ns):
        if par[s] >= 0:
            kids[par[s]].append(s)
    for s in range(ns):
        p = par[s]
        if p >= 0:
            assert sy.level[p] > sy.level[s]
            assert is_top[p] or owner[p] == owner[s]             # subtrees are closed downwards: one owner
            if is_top[s]:
                assert is_top[p]                                 # the top is closed upwards
        if is_top[s]:
            assert sy.level[s] >= L0 and kids[s]
            assert owner[s] in {owner[d] for d in kids[s]}       # owned inside the group of ranks it joins
        else:
            assert sy.level[s] < L0
    cross = sorted(d for d in range(ns) if par[d] >= 0 and owner[par[d]] != owner[d])
    assert sorted(E["child"].tolist()) == cross and len(cross) == info["n_edges"]
    for i, d in enumerate(E["child"]):
        assert E["src"][i] == owner[d] and E["dst"][i] == owner[par[d]] and E["level"][i] == sy.level[par[d]] >= L0
        assert E["cb_count"][i] == (sy.row_ptr[d + 1] - sy.row_ptr[d] - (sy.super_first[d + 1] - sy.super_first[d])) ** 2
    assert (np.diff(E["level"]) >= 0).all()
    assert sy.level.max() + 1 == L0 + K
    # ---- sharded host walk -----------------------------------------------------------------------
    sim = HostSim(sy, n, np.asarray(Q.data))
    mine = lambda s: owner[s] == rank
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

    def exchange(store, lev, width, tagbase):
        """the cross-rank edges whose parent sits at level `lev`: src -> dst"""
        for i in np.flatnonzero(E["level"] == lev):
            d, src, dst = int(E["child"][i]), int(E["src"][i]), int(E["dst"][i])
            m = int(sim.r[d] - sim.c[d])
            if rank == src:
                dist.send(torch.from_numpy(np.ascontiguousarray(store.pop(d))), dst=dst, tag=tagbase + i)
            elif rank == dst:
                buf = torch.empty((m, m if width is None else width), dtype=torch.float64)
                dist.recv(buf, src=src, tag=tagbase + i)
                store[d] = buf.numpy()

    for s in sim.order:                                   # phase 0: own subtrees (levels < shard_level)
        if not is_top[s] and mine(s):
            factor_front(s)
    for k in range(K):                                    # top levels, one phase each
        exchange(cb, L0 + k, None, 0)
        for s in sim.order:
            if is_top[s] and sy.level[s] == L0 + k and mine(s):
                factor_front(s)
    # ---- sharded solve ---------------------------------------------------------------------------
    k_rhs = 3
    Bp = np.random.default_rng(5).standard_normal((n, k_rhs))      # right-hand sides in elimination order
    X = Bp.copy()
    W = {}

    def fwd_front(s):
        c, r = sim.c[s], sim.r[s]
        f = np.zeros((r, k_rhs)); rows = sim.rows(s)
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
        if not is_top[s] and mine(s):
            fwd_front(s)
    for k in range(K):
        exchange(W, L0 + k, k_rhs, 1000)
        for s in sim.order:
            if is_top[s] and sy.level[s] == L0 + k and mine(s):
                fwd_front(s)
    for k in range(K - 1, -1, -1):
        lev_fronts = [s for s in sim.order if is_top[s] and sy.level[s] == L0 + k]
        for s in lev_fronts[::-1]:
            if mine(s):
                bwd_front(s)
        for s in lev_fronts:                              # x of the level's fronts: owner -> all
            rows = sim.rows(s)[:sim.c[s]]
            buf = torch.from_numpy(np.ascontiguousarray(X[rows]))
            dist.broadcast(buf, src=int(owner[s]))
            X[rows] = buf.numpy()
    for s in sim.order[::-1]:
        if not is_top[s] and mine(s):
            bwd_front(s)
    for s in range(ns):                                   # owned x -> rank 0 (per front here; per subtree on the device)
        if is_top[s] or owner[s] == 0:
            continue
        rows = sim.rows(s)[:sim.c[s]]
        if rank == owner[s]:
            dist.send(torch.from_numpy(np.ascontiguousarray(X[rows])), dst=0, tag=2000 + s)
        elif rank == 0:
            buf = torch.empty((len(rows), k_rhs), dtype=torch.float64)
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
        q.put((float(t.item()), ref, dense, info, [int(is_top.sum())] + [int((owner == kk).sum()) for kk in range(world)], solve_err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])
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


def test_plan_summary_time_bounds_from_level_times():
    """gmrfx/shard.py plan_summary with measured per-level times (here: a synthetic profile): both time bounds lie between 1
    and the world size, the latency form (top levels not shortened) below the share form, and with times proportional to the
    flops of a level the share form cannot beat the flop count of the heaviest rank."""
    import gmrfx
    from gmrfx import shard, spde
    mesh = spde.grid_mesh_2d(90, 90, jitter=0.25, seed=4)
    Q = spde.matern_precision(mesh, 0, 0.2)
    for W in (2, 4):
        be = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True, shard_rank=0, shard_world=W)
        sy = be.symbolic()
        c = np.diff(sy.super_first).astype(float); m = np.diff(sy.row_ptr).astype(float) - c
        fl = c ** 3 / 3 + c * c * m + c * m * m
        par = np.asarray(sy.super_parent); h = np.zeros(len(c), int)
        for s in range(len(c)):
            if par[s] >= 0:
                h[par[s]] = max(h[par[s]], h[s] + 1)
        level_ms = np.concatenate([[0.0], [fl[h == l].sum() * 1e-9 for l in range(h.max() + 1)]])
        p = shard.plan_summary(be, level_ms)
        assert abs(p["measured_ms_one_gpu"] - level_ms.sum()) < 1e-9 * level_ms.sum()
        assert 1.0 <= p["time_bound_speedup_latency"] <= p["time_bound_speedup_share"] <= W + 1e-9
        owner, top = be.shard_owner(with_top=True)
        heaviest = max(fl[owner == k].sum() for k in range(W))
        assert p["time_bound_speedup_share"] <= fl.sum() / heaviest + 1e-9
        assert p["top_levels"] and len(p["top_levels"]) == len(p["top_levels_ms"])
        with pytest.raises(ValueError):
            shard.plan_summary(be, level_ms[:3])
        be.close()


def _dist_root_worker(rank, world, port, q):
    """Host walk (numpy) of the DISTRIBUTED ROOT protocol on the structures the library exports (gmrfx_shard_dist_root):
    column ranges of the children's contribution blocks to the owners of the 256-column blocks, assembly of the own blocks,
    per block: owner factors its block column, broadcast, everybody updates its own later blocks."""
    for p in (os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GMRFX_DIST_ROOT_MIN"] = "512"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scipy.sparse as sp
    import gmrfx
    from mf_hostsim import HostSim
    # two independent 9 x 9 grid Laplacians, both coupled to one dense 700-column block ordered last (natural ordering):
    # the dense block is the root front (3 outer blocks of 256 columns), the two grids are its two child subtrees
    g = 9
    A1 = sp.diags([-np.ones(g - 1), 2.0 * np.ones(g), -np.ones(g - 1)], [-1, 0, 1])
    G = sp.kron(sp.identity(g), A1) + sp.kron(A1, sp.identity(g)) + 0.5 * sp.identity(g * g)
    nd = 700
    rng = np.random.default_rng(11)
    Dm = rng.standard_normal((nd, nd)); Dm = Dm @ Dm.T / nd + 5.0 * np.eye(nd)
    H1 = sp.random(g * g, nd, density=0.05, random_state=1) * 0.1
    H2 = sp.random(g * g, nd, density=0.05, random_state=2) * 0.1
    Q = sp.bmat([[G, None, H1], [None, G, H2], [H1.T, H2.T, sp.csr_matrix(Dm)]], format="csc")
    Q.sort_indices()
    n = Q.shape[0]
    be = gmrfx.MI355XBackend(Q, ordering=np.arange(n), symbolic_only=True, shard_rank=rank, shard_world=world)
    sy = be.symbolic()
    owner, is_top = be.shard_owner(with_top=True)
    info, E, dr = be.shard_info(), be.shard_edges(), be.shard_dist_root()
    L0, K = info["shard_level"], info["n_top_levels"]
    R = dr["root"]
    assert R >= 0 and dr["cols"] >= nd and dr["blocks"] == (dr["cols"] + 255) // 256 >= 3 and dr["world"] == world   # (amalgamation may add grid columns)
    sim = HostSim(sy, n, np.asarray(Q.data))
    ns = sim.ns
    assert sy.super_parent[R] == -1 and sy.level[R] == L0 + K - 1 and (sy.level == sy.level[R]).sum() == 1
    mine = lambda s: owner[s] == rank
    cb = {}

    def factor_front(s):
        c, r = sim.c[s], sim.r[s]
        F = np.zeros((r, r)); P = sim.panel(sim.L, s); F[:, :c] = P
        for d in sim.children[s]:
            rel = sim.rel(d)
            F[np.ix_(rel, rel)] += cb.pop(d)
        F = np.tril(F); F = F + np.tril(F, -1).T
        L11 = np.linalg.cholesky(F[:c, :c]); L21 = np.linalg.solve(L11, F[c:, :c].T).T
        P[:c, :] = np.tril(L11); P[c:, :] = L21
        cb[s] = F[c:, c:] - L21 @ L21.T

    root_kids = set(int(d) for d in dr["child"])
    assert root_kids == set(sim.children[R])
    for lev in range(int(sy.level.max()) + 1):
        for i in np.flatnonzero(E["level"] == lev):                     # whole blocks along the other cross edges
            d, src, dst = int(E["child"][i]), int(E["src"][i]), int(E["dst"][i])
            if d in root_kids:
                continue
            m = int(sim.r[d] - sim.c[d])
            if rank == src:
                dist.send(torch.from_numpy(np.ascontiguousarray(cb.pop(d))), dst=dst, tag=i)
            elif rank == dst:
                buf = torch.empty((m, m), dtype=torch.float64); dist.recv(buf, src=src, tag=i); cb[d] = buf.numpy()
        for s in sim.order:
            if sy.level[s] == lev and mine(s) and s != R:
                factor_front(s)
    # ---- the distributed root ---------------------------------------------------------------------------------
    c = int(sim.c[R]); P = sim.panel(sim.L, R)
    assert P.shape == (c, c)
    parts = {}                      # (child, block) -> (k0, columns of the child's block)
    seen = {d: 0 for d in root_kids}
    for k in range(len(dr["child"])):
        d, b, cnt = int(dr["child"][k]), int(dr["block"][k]), int(dr["count"][k])
        md = int(sim.r[d] - sim.c[d]); k0 = seen[d]; w = cnt // md
        assert cnt == w * md and (sim.rel(d)[k0:k0 + w] // 256 == b).all()
        seen[d] += w
        src, dst = int(owner[d]), b % world
        if src == dst:
            if rank == src:
                parts[(d, b)] = (k0, cb[d][:, k0:k0 + w].copy())
        elif rank == src:
            dist.send(torch.from_numpy(np.ascontiguousarray(cb[d][:, k0:k0 + w])), dst=dst, tag=5000 + k)
        elif rank == dst:
            buf = torch.empty((md, w), dtype=torch.float64); dist.recv(buf, src=src, tag=5000 + k)
            parts[(d, b)] = (k0, buf.numpy())
    for d in root_kids:
        assert seen[d] == sim.r[d] - sim.c[d]                             # every column of every child exactly once
    for (d, b), (k0, cols) in parts.items():                              # assemble my blocks (lower triangle)
        rel = sim.rel(d)
        for jj in range(cols.shape[1]):
            kk = k0 + jj
            P[rel[kk:], rel[kk]] += cols[kk:, jj]
    nb = dr["blocks"]
    for b in range(nb):
        lo, hi = 256 * b, min(256 * b + 256, c)
        if b % world == rank:
            A = np.tril(P[lo:hi, lo:hi]); A = A + np.tril(A, -1).T
            Lbb = np.linalg.cholesky(A)
            P[lo:hi, lo:hi] = Lbb
            P[hi:, lo:hi] = np.linalg.solve(Lbb, P[hi:, lo:hi].T).T
        buf = torch.from_numpy(np.ascontiguousarray(P[:, lo:hi]))
        dist.broadcast(buf, src=b % world)
        P[:, lo:hi] = buf.numpy()
        for j in range(b + 1, nb):
            if j % world == rank:
                jl, jh = 256 * j, min(256 * j + 256, c)
                P[jl:, jl:jh] -= P[jl:, lo:hi] @ P[jl:jh, lo:hi].T
    # every rank now holds the whole root factor: against dense LAPACK on the permuted matrix
    perm = be.ordering_permutation()
    Lref = np.linalg.cholesky(Q.toarray()[np.ix_(perm, perm)])
    first = int(sy.super_first[R])
    err = float(np.abs(np.tril(P) - Lref[first:first + c, first:first + c]).max())
    part = 2.0 * sum(np.log(np.diag(sim.panel(sim.L, s)[:sim.c[s]])).sum() for s in range(ns) if mine(s))
    t = torch.tensor([part], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    q.put((rank, err, float(t.item()), float(2.0 * np.log(np.diag(Lref)).sum()), int(owner[R])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_root_front_host_walk_gloo(world):
    """the distributed dense root front on the host walk (numpy blocks, gloo) against dense LAPACK --
    the column-range lists and block ownership the library exports drive the exchange; every rank ends with the whole root
    factor, equal to the dense Cholesky factor's block, and the all-reduced log-determinant is the dense one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_root_worker, args=(r, world, 29850 + world, q)) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, err, ld, ld_ref, root_owner in got:
        assert err < 1e-10, (rank, err)
        assert abs(ld - ld_ref) < 1e-10 * abs(ld_ref)
