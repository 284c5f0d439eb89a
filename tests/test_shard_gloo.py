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


@pytest.mark.parametrize("world", [2, 3, 5, 8])
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
        par = np.asarray(sy.super_parent); hg = np.zeros(len(c), int)
        for s in range(len(c)):
            if par[s] >= 0:
                hg[par[s]] = max(hg[par[s]], hg[s] + 1)
        h = np.zeros(len(c), int)                   # the unsharded schedule's levels: depth below the root, counted down from the height
        for s in range(len(c) - 1, -1, -1):
            h[s] = hg.max() if par[s] < 0 else h[par[s]] - 1
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


def _dist_fronts_worker(rank, world, port, q, kind="dense"):
    """Host walk (numpy) of the DISTRIBUTED TOP FRONT protocol on the structures the library exports (gmrfx_shard_dist_fronts,
    gmrfx_shard_transfers): column ranges of the children's contribution blocks to the owners of the 256-column blocks they
    fall into, assembly of the own panel blocks, per panel block: owner factors its block column, broadcast inside the group,
    every member updates its own later blocks; then every member's own column blocks of the contribution block."""
    for p in (os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"), os.path.join(ROOT, "oracle"), HERE):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GMRFX_DIST_MIN"] = "256"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scipy.sparse as sp
    import gmrfx
    from mf_hostsim import HostSim
    # four independent 9 x 9 grid Laplacians; grids 1, 2 coupled to a dense 300-column block M1, grids 3, 4 to M2, both M's to a
    # dense 700-column block ordered last (natural ordering): the root front (3 blocks of 256 columns) has the two M fronts
    # (2 panel blocks + a 700-row contribution block each) as children, each M front two grid subtrees
    g = 9
    A1 = sp.diags([-np.ones(g - 1), 2.0 * np.ones(g), -np.ones(g - 1)], [-1, 0, 1])
    G = sp.kron(sp.identity(g), A1) + sp.kron(A1, sp.identity(g)) + 0.5 * sp.identity(g * g)
    nm, nd = 300, 700
    rng = np.random.default_rng(11)
    dense = lambda k: (lambda D: sp.csr_matrix(D @ D.T / k + 5.0 * np.eye(k)))(rng.standard_normal((k, k)))
    H = [sp.random(g * g, nm, density=0.05, random_state=1 + i) * 0.1 for i in range(4)]
    C = [sp.csr_matrix(rng.standard_normal((nm, nd)) * 0.02) for _ in range(2)]
    N = None
    Q = sp.bmat([[G, N, H[0], N, N, N, N], [N, G, H[1], N, N, N, N], [H[0].T, H[1].T, dense(nm), N, N, N, C[0]],
                 [N, N, N, G, N, H[2], N], [N, N, N, N, G, H[3], N], [N, N, N, H[2].T, H[3].T, dense(nm), C[1]],
                 [N, N, C[0].T, N, N, C[1].T, dense(nd)]], format="csc")
    Q.sort_indices()
    n = Q.shape[0]
    if kind == "spacetime":
        # SURVEY 8 f3 "distributed separators": the space-time posterior kron(Q_t, Q_s) + diag(h) (separable.jl:143-172) on the
        # space-time nested dissection -- its top separators (time slabs / space cuts, > 256 columns) are distributed fronts
        from gmrfx import spde, spacetime
        ms = spde.grid_mesh_2d(18, 17, jitter=0.2, seed=4)
        Qs = spde.matern_precision(ms, 0, 0.4)
        T = 16
        Q = spacetime.spacetime_precision(spde.ar1_precision(T, 0.8), Qs, obs_diag=np.random.default_rng(5).uniform(0.5, 2.0, T * Qs.shape[0]))
        n = Q.shape[0]
        be = gmrfx.MI355XBackend(Q, coords=spacetime.spacetime_coords(ms.points, T), symbolic_only=True, shard_rank=rank, shard_world=world)
    else:
        be = gmrfx.MI355XBackend(Q, ordering=np.arange(n), symbolic_only=True, shard_rank=rank, shard_world=world)
    sy = be.symbolic()
    owner, is_top = be.shard_owner(with_top=True)
    info, df, X = be.shard_info(), be.shard_dist_fronts(), be.shard_transfers()
    L0, K = info["shard_level"], info["n_top_levels"]
    # the host walk keeps EVERY panel whole in its own numpy storage (round 6: the library stores a distributed front without
    # trailing rows block-cyclically on the members that do not own it): storage from an unsharded analysis of the same ordering,
    # plan (owners, levels, groups, transfers) from this rank's sharded one
    sy_full = gmrfx.MI355XBackend(Q, ordering=be.ordering_permutation(), symbolic_only=True).symbolic()
    assert np.array_equal(sy_full.super_first, sy.super_first) and np.array_equal(sy_full.row_ptr, sy.row_ptr)
    sim = HostSim(sy_full, n, np.asarray(Q.data))
    ns = sim.ns
    dfi = {int(s): i for i, s in enumerate(df["front"])}
    R = int(np.flatnonzero(sy.super_parent == -1)[-1])
    if kind == "spacetime":
        assert len(dfi) >= 1 and max(df["cols"]) > 256
    else:
        assert R in dfi and df["cols"][dfi[R]] >= nd and df["group"][dfi[R]] == list(range(world))
        if world >= 4:      # an M front below the root is distributed as well, by a sub-group (the other may be amalgamated into the root)
            assert any(int(s) != R and len(df["group"][i]) < world and sim.r[int(s)] > sim.c[int(s)] for s, i in dfi.items())
    groups = {tuple(G_): None for G_ in df["group"]}
    for key in groups:
        groups[key] = None if len(key) == world else dist.new_group(ranks=list(key))
    nbp = lambda s: (int(sim.c[s]) + 255) // 256
    panel_owner = lambda s, b: df["group"][dfi[s]][b % len(df["group"][dfi[s]])] if s in dfi else int(owner[s])
    cb_owner = lambda s, qb: df["group"][dfi[s]][(nbp(s) + qb) % len(df["group"][dfi[s]])] if s in dfi else int(owner[s])
    need = lambda p, j: panel_owner(p, j // 256) if j < sim.c[p] else cb_owner(p, (j - int(sim.c[p])) // 256)
    cb, have = {}, {}

    def slot(d):
        if d not in cb:
            md = int(sim.r[d] - sim.c[d]); cb[d] = np.zeros((md, md)); have[d] = np.zeros(md, bool)
        return cb[d]

    def factor_front(s):
        c, r = sim.c[s], sim.r[s]
        F = np.zeros((r, r)); P = sim.panel(sim.L, s); F[:, :c] = P
        for d in sim.children[s]:
            assert have[d].all(), (s, d)                # every column of the child is here (own or received)
            rel = sim.rel(d)
            F[np.ix_(rel, rel)] += np.tril(cb.pop(d))
        F = np.tril(F); F = F + np.tril(F, -1).T
        L11 = np.linalg.cholesky(F[:c, :c]); L21 = np.linalg.solve(L11, F[c:, :c].T).T
        P[:c, :] = np.tril(L11); P[c:, :] = L21
        slot(s)[:] = F[c:, c:] - L21 @ L21.T; have[s][:] = True

    def factor_dist_front(s):
        G_ = df["group"][dfi[s]]; pg = groups[tuple(G_)]
        c, r = int(sim.c[s]), int(sim.r[s]); m = r - c
        P = sim.panel(sim.L, s); Cb = slot(s)
        for d in sim.children[s]:                       # assemble my panel / contribution-block columns
            rel = sim.rel(d)
            for k in range(len(rel)):
                j = int(rel[k])
                if need(s, j) != rank:
                    continue
                assert have[d][k], (s, d, k)
                if j < c:
                    P[rel[k:], j] += cb[d][k:, k]
                else:
                    Cb[rel[k:] - c, j - c] += cb[d][k:, k]
        nb = nbp(s)
        for b in range(nb):
            lo, hi = 256 * b, min(256 * b + 256, c)
            src = panel_owner(s, b)
            if src == rank:
                A = np.tril(P[lo:hi, lo:hi]); A = A + np.tril(A, -1).T
                Lbb = np.linalg.cholesky(A)
                P[lo:hi, lo:hi] = Lbb
                P[hi:, lo:hi] = np.linalg.solve(Lbb, P[hi:, lo:hi].T).T
            buf = torch.from_numpy(np.ascontiguousarray(P[:, lo:hi]))
            dist.broadcast(buf, src=src, group=pg)
            P[:, lo:hi] = buf.numpy()
            for j in range(b + 1, nb):
                if panel_owner(s, j) == rank:
                    jl, jh = 256 * j, min(256 * j + 256, c)
                    P[jl:, jl:jh] -= P[jl:, lo:hi] @ P[jl:jh, lo:hi].T
        L21 = P[c:, :]
        for qb in range((m + 255) // 256):
            if cb_owner(s, qb) == rank:
                ql, qh = 256 * qb, min(256 * qb + 256, m)
                Cb[ql:, ql:qh] -= L21[ql:] @ L21[ql:qh].T
                have[s][ql:qh] = True

    for lev in range(int(sy.level.max()) + 1):
        for i in np.flatnonzero(X["level"] == lev):
            d, src, dst, k0, cnt = (int(X[nm_][i]) for nm_ in ("child", "src", "dst", "col0", "count"))
            md = int(sim.r[d] - sim.c[d]); w = cnt // md
            assert cnt == w * md and src != dst
            p = int(sy.super_parent[d])
            assert all(cb_owner(d, k // 256) == src and need(p, int(sim.rel(d)[k])) == dst for k in range(k0, k0 + w))
            if rank == src:
                assert have[d][k0:k0 + w].all()
                dist.send(torch.from_numpy(np.ascontiguousarray(cb[d][:, k0:k0 + w])), dst=dst, tag=int(i))
            elif rank == dst:
                buf = torch.empty((md, w), dtype=torch.float64); dist.recv(buf, src=src, tag=int(i))
                slot(d)[:, k0:k0 + w] = buf.numpy(); have[d][k0:k0 + w] = True
        for s in sim.order:
            if sy.level[s] != lev:
                continue
            if int(s) in dfi:
                if rank in df["group"][dfi[int(s)]]:
                    factor_dist_front(int(s))
            elif owner[s] == rank:
                factor_front(int(s))
    # every panel this rank holds (own fronts; distributed fronts of its groups, complete after the last broadcast) against dense
    # LAPACK on the permuted matrix
    perm = be.ordering_permutation()
    Lref = np.linalg.cholesky(Q.toarray()[np.ix_(perm, perm)])
    err, held = 0.0, 0
    for s in range(ns):
        if owner[s] == rank or (s in dfi and rank in df["group"][dfi[s]]):
            c = int(sim.c[s]); first = int(sy.super_first[s]); P = sim.panel(sim.L, s).copy()
            P[:c, :] = np.tril(P[:c, :])
            err = max(err, float(np.abs(P - Lref[np.ix_(sim.rows(s), np.arange(first, first + c))]).max()))
            held += 1
    part = 2.0 * sum(np.log(np.diag(sim.panel(sim.L, s)[:sim.c[s]])).sum() for s in range(ns) if owner[s] == rank)
    t = torch.tensor([part], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    q.put((rank, err, float(t.item()), float(2.0 * np.log(np.diag(Lref)).sum()), len(dfi), held))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_top_fronts_host_walk_gloo(world):
    """VERDICT r2 next #4, generalised: the distributed dense top fronts (root AND the fronts below it, with contribution blocks)
    on the host walk (numpy blocks, gloo) against dense LAPACK -- the column-range transfer list, the groups and the block
    ownership the library exports drive the exchange; every member of a group ends with the front's whole panel, equal to the
    dense Cholesky factor's block, and the all-reduced log-determinant is the dense one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_fronts_worker, args=(r, world, 29850 + world, q)) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=240) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, err, ld, ld_ref, ndist, held in got:
        assert err < 1e-10, (rank, err)
        assert abs(ld - ld_ref) < 1e-10 * abs(ld_ref)
        assert ndist >= (2 if world >= 4 else 1) and held > 0


@pytest.mark.parametrize("world", [2])          # (world 4: the one-GPU rehearsal, tests/test_gpu_parity.py::test_sharded_spacetime_rehearsal_on_one_gpu)
def test_distributed_separators_of_a_spacetime_precision_host_walk_gloo(world):
    """SURVEY 8 f3, second half ("distributed separators"): the same host walk on the SPACE-TIME posterior precision
    kron(AR(1), Matern) + diag(h) with the space-time nested dissection -- the separators at the top of its tree are distributed
    fronts; panels against dense LAPACK on the permuted matrix, all-reduced log-determinant."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_fronts_worker, args=(r, world, 29950 + world, q, "spacetime")) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, err, ld, ld_ref, ndist, held in got:
        assert err < 1e-10, (rank, err)
        assert abs(ld - ld_ref) < 1e-10 * abs(ld_ref)
        assert ndist >= 1 and held > 0
