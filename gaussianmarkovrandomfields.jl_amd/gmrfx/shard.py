"""One factorisation sharded over the ranks of a torch.distributed process group (one GPU each).

SURVEY section 8(e): the supernodal tree is cut top-down into subtrees that are dealt to the ranks; every front above
them (the "top") is owned by ONE rank of the group whose subtrees it joins (`Symbolic::owner`, csrc/symbolic.cpp), so
independent top fronts run on different GPUs. Data crosses ranks only along tree edges whose ends have different owners
-- the Schur-complement contribution block of the child in the factorisation, its update vector in the forward sweep --
and goes point-to-point, src -> dst, into the same offset of the destination's buffers (all ranks share one layout);
after its backward step the owner of a top front broadcasts the front's x. log det Q is an all-reduce of partial sums,
a non-positive pivot an all-reduce (min). The top levels run one level per phase; all transfers of a phase are posted
as ONE batch (batch_isend_irecv). This is the exchange the reference's CPU solver (CHOLMOD behind
src/workspace/backend.jl:165-189) never needs: it has one address space.

Device buffers are handed to torch.distributed without copies: the library's arena / X / W buffers are wrapped as torch
tensors through __cuda_array_interface__ (RCCL on MI355X nodes); the gloo rehearsal stages through host tensors.

Ordering (round 3): the handle runs on TORCH'S CURRENT STREAM (gmrfx_set_stream) with asynchronous phases, so a phase, the
transfers behind it and the next phase are ordered by the stream itself -- torch.distributed makes its communication stream
wait for the current stream when an operation is enqueued and the current stream wait for the operation in `wait()` --
and the host never blocks between them (round 2: a stream synchronisation inside every phase call plus a device-wide
synchronisation after every exchange, 2K + 2 host round trips per step). The per-level views / transfer lists are built
once. Every rank keeps only its own part of the factor (panels, arena, update vectors: csrc/symbolic.cpp)."""
from __future__ import annotations

import os

import numpy as np


class _DevView:
    """Zero-copy view of library-owned device memory for torch.as_tensor."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class ShardedFactor:
    def __init__(self, Q, dist, device: int = 0, coords=None, **kw):
        import torch
        from .backend import MI355XBackend
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dev = torch.device("cuda", device)
        self.be = MI355XBackend(Q, coords=coords, device=device, factorize=False, shard_rank=self.rank,
                                shard_world=self.world, **kw)
        self.info = self.be.shard_info()
        self.K = self.info["n_top_levels"]
        self.L0 = self.info["shard_level"]
        self.edges = self.be.shard_edges()
        self.top_rows = self.be.shard_rows(2)
        self.sub_rows = self.be.shard_rows(3)
        self.host_staging = dist.get_backend() == "gloo"      # rehearsal: gloo moves host tensors only
        self.last_info = 0
        self._info_pending = False
        self._df_views = {}
        self._p2p_cache = {}                      # prebuilt P2POp lists of the recurring transfer lists (see _p2p)
        # one stream orders everything: the library's kernels, torch's copies, the transfers torch.distributed enqueues
        self.be.set_stream(torch.cuda.current_stream(self.dev).cuda_stream, True, True)
        # per top level: the cross-edge transfers of the factorisation (contribution blocks) and of the forward sweep
        # (update vectors, per right-hand side), built once
        e = self.edges
        self._cb_items, self._w_items = [], []
        for k in range(self.K):
            idx = np.flatnonzero(e["level"] == self.L0 + k)
            self._cb_items.append([(int(e["src"][i]), int(e["dst"][i]), 0, int(e["cb_offset"][i]), int(e["cb_count"][i])) for i in idx])
            self._w_items.append([(int(e["src"][i]), int(e["dst"][i]), int(e["w_row0"][i]), int(e["w_nrows"][i])) for i in idx])
        owner, r0, nr, lv = self.top_rows
        self._top_blocks = [[(int(owner[i]), int(r0[i]), int(nr[i])) for i in np.flatnonzero(lv == self.L0 + k)] for k in range(self.K)]
        so, sr0, snr, _ = self.sub_rows
        self._sub_blocks = [(int(so[i]), int(sr0[i]), int(snr[i])) for i in range(len(so))]
        # contribution-block transfers of the factorisation (column ranges; csrc/symbolic.h xf_*), per top level
        x = self.be.shard_transfers()
        self._cb_items = [[(int(x["src"][i]), int(x["dst"][i]), 0, int(x["offset"][i]), int(x["count"][i]))
                           for i in np.flatnonzero(x["level"] == self.L0 + k)] for k in range(self.K)]
        # distributed top fronts (csrc/symbolic.h): factored by their whole group, 256-column blocks dealt cyclically; one
        # process group per distinct group of ranks (created collectively, same order everywhere)
        self.df = self.be.shard_dist_fronts()
        self._dist_of_level = [[i for i in range(len(self.df["front"])) if int(self.df["level"][i]) == self.L0 + k] for k in range(self.K)]
        self._pg = {}
        for g in self.df["group"]:
            key = tuple(g)
            if key not in self._pg:
                self._pg[key] = None if len(g) == self.world else dist.new_group(ranks=list(g))

    # ---- transfers ----------------------------------------------------------------------------------
    def _view(self, which: int, off: int, cnt: int):
        base = self.be.device_ptr(which)
        return self.torch.as_tensor(_DevView(base + 8 * int(off), int(cnt)), device=self.dev)

    def _p2p(self, items, key=None):
        """items: (src, dst, which buffer, offset, count) -- every rank passes the same list; the transfers this rank
        takes part in are posted as one batch. `key` names a list that recurs every step (the contribution blocks / update
        vectors of one top level, the owned rows going home): its views of the library's buffers and its P2POp objects are
        then built ONCE and posted again as they are (round 4 rebuilt both on every call: a few hundred Python objects per
        step on the critical path between two phases). Device path only: the gloo rehearsal stages through fresh host copies."""
        t, dist = self.torch, self.dist
        if key is not None and not self.host_staging:
            ops = self._p2p_cache.get(key)
            if ops is None:
                ops = []
                for tag, (src, dst, which, off, cnt) in enumerate(items):
                    if src == dst or self.rank not in (src, dst) or cnt == 0:
                        continue
                    v = self._view(which, off, cnt)
                    ops.append(dist.P2POp(dist.isend, v, dst, tag=tag) if self.rank == src else dist.P2POp(dist.irecv, v, src, tag=tag))
                self._p2p_cache[key] = ops
            if ops:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()            # (RCCL: the current STREAM waits, the host does not)
            return
        ops, post = [], []
        for tag, (src, dst, which, off, cnt) in enumerate(items):
            if src == dst or self.rank not in (src, dst) or cnt == 0:
                continue
            v = self._view(which, off, cnt)
            if self.rank == src:
                ops.append(dist.P2POp(dist.isend, v.cpu() if self.host_staging else v, dst, tag=tag))
            elif self.host_staging:
                buf = t.empty(int(cnt), dtype=t.float64)
                ops.append(dist.P2POp(dist.irecv, buf, src, tag=tag))
                post.append((v, buf))
            else:
                ops.append(dist.P2POp(dist.irecv, v, src, tag=tag))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()                # (RCCL: the current STREAM waits; gloo: the host does)
        for v, buf in post:
            v.copy_(buf)                # host -> device on the current stream: ordered before the next phase

    def _bcast_rows(self, blocks, nrhs: int):
        """blocks: (owner, first row, rows) of X row blocks; every owner broadcasts its blocks IN PLACE -- the views of the
        library's X buffer themselves, posted asynchronously and waited for together: no fused staging buffer, no copies back
        (round 3 concatenated an owner's blocks into one buffer and copied them out again on every rank)."""
        dist = self.dist
        if self.host_staging:                      # gloo rehearsal: host tensors only
            for o, r0, nr in blocks:
                v = self._view(2, int(r0) * nrhs, int(nr) * nrhs)
                buf = v.cpu()
                dist.broadcast(buf, src=int(o))
                if self.rank != int(o):
                    v.copy_(buf)
            return
        works = [dist.broadcast(self._view(2, int(r0) * nrhs, int(nr) * nrhs), src=int(o), async_op=True) for o, r0, nr in blocks]
        for w in works:
            w.wait()                               # (RCCL: the current STREAM waits, the host does not)

    def _edges_of_level(self, lev: int):
        e = self.edges
        return np.flatnonzero(e["level"] == lev)

    # ---- factorisation ------------------------------------------------------------------------------
    def refactorize_dev(self, d_nzval_ptr: int, check: bool = True) -> int:
        """check = False leaves the pivot report (a host round trip: it would sit between the factorisation and the solve behind
        it) to the next logdet() call, which needs the host anyway; `last_info` is then valid after that call."""
        be = self.be
        be.refactorize_phase_dev(d_nzval_ptr, 0)
        for k in range(self.K):
            self._p2p(self._cb_items[k], key=("cb", k))
            for i in self._dist_of_level[k]:
                if self.rank in self.df["group"][i]:
                    self._factor_distributed_front(d_nzval_ptr, i)
            be.refactorize_phase_dev(d_nzval_ptr, 1 + k)
        self._info_pending = True
        return self._reduce_info() if check else 0

    def _reduce_info(self) -> int:
        """first non-positive pivot over all ranks (0 = none), like the `info` of gmrfx_refactorize"""
        t = self.torch
        fc = int(self.be.stats()["fail_col"])
        v = t.tensor([fc if fc >= 0 else 2 ** 62], dtype=t.int64, device="cpu" if self.host_staging else self.dev)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.MIN)
        self.last_info = 0 if int(v.item()) >= 2 ** 62 else int(v.item()) + 1
        self._info_pending = False
        return self.last_info

    def _bcast_block(self, v, src: int, pg, async_: bool):
        """broadcast of one panel block inside the front's group; returns a function that waits for it"""
        if self.host_staging:
            buf = v.cpu()
            self.dist.broadcast(buf, src=src, group=pg)
            if self.rank != src:
                v.copy_(buf)
            return lambda: None
        if not async_:
            self.dist.broadcast(v, src=src, group=pg)
            return lambda: None
        w = self.dist.broadcast(v, src=src, group=pg, async_op=True)
        return w.wait

    def _factor_distributed_front(self, d_nzval_ptr: int, i: int) -> None:
        """A top front factored by its group: per 256-column panel block its owner factors the block column and broadcasts it
        inside the group (whole columns: one contiguous piece of the panel where a member stores it whole, of its window of two
        received blocks where it keeps only its own blocks -- gmrfx_dist_front_block), every member updates its own later blocks; then every member computes its own column blocks of the contribution block. The children's blocks have
        arrived as column ranges at the owners of the blocks they fall into (the level's transfers).
        LOOK-AHEAD (round 4): once block b has arrived, the owner of block b + 1 applies b to that block FIRST (phase 4), factors
        it and posts its broadcast asynchronously; everybody applies b to the rest of its blocks (phase 5) while that broadcast is
        in flight. Per entry the same sums in the same order as the strictly sequential loop (factor -> broadcast -> update) of
        round 3: bit-identical factor (tests/test_gpu_parity.py::test_distributed_top_fronts_rehearsal_on_one_gpu)."""
        be, df = self.be, self.df
        s, c, ld, G = int(df["front"][i]), int(df["cols"][i]), int(df["panel_ld"][i]), df["group"][i]
        pg = self._pg[tuple(G)]
        nb = (c + 255) // 256
        views = self._df_views.get(i)
        if views is None:                                           # built once per front: where THIS rank keeps block b (the whole
            views = [self._view(1, *be.dist_front_block(s, b)) for b in range(nb)]     # panel, or own blocks + a two-block window)
            self._df_views[i] = views
        be.dist_front_phase(d_nzval_ptr, s, 0)                      # assemble my panel blocks
        be.dist_front_phase(d_nzval_ptr, s, 1, 0)                   # its owner factors block 0
        wait = self._bcast_block(views[0], G[0], pg, async_=False)
        for b in range(nb):
            wait()                                                  # block b is complete on every member
            if b + 1 < nb:
                be.dist_front_phase(d_nzval_ptr, s, 4, b)           # b -> block b + 1 (its owner only)
                be.dist_front_phase(d_nzval_ptr, s, 1, b + 1)       # its owner factors block b + 1 ...
                wait = self._bcast_block(views[b + 1], G[(b + 1) % len(G)], pg, async_=True)     # ... and sends it off
                be.dist_front_phase(d_nzval_ptr, s, 5, b)           # b -> my other later blocks, beside that broadcast
        be.dist_front_phase(d_nzval_ptr, s, 3)                      # my blocks of the contribution block

    # ---- solve --------------------------------------------------------------------------------------
    def solve_dev(self, d_B: int, ldb: int, nrhs: int, d_X: int, ldx: int, gather: bool = True) -> None:
        """Q X = B with the factor sharded over the ranks; B column-major n x nrhs, of which this rank reads the rows `needed_rows()`
        only (its subtrees' and its own top fronts': B may be row-sharded, the rest left unset). gather = True: X (full) is
        produced on rank 0 -- every rank sends the rows of its subtrees home, 7/8 of n x nrhs doubles into ONE rank's links at world
        8. gather = False: X stays DISTRIBUTED -- every rank writes its own d_X, of which the rows `valid_rows()` names are final
        (its subtrees' and every top front's, which are broadcast anyway): no transfer at all behind the backward sweep; a caller
        that wants all of X on one rank gathers those rows itself, later or never (a mean / a sample that is reduced further
        stays where it is). Any number of right-hand sides: passes of up to 64 columns."""
        for j0 in range(0, int(nrhs), 64):
            self._solve_pass(d_B + 8 * j0 * ldb, ldb, min(64, nrhs - j0), d_X + 8 * j0 * ldx, ldx, backward_only=False, gather=gather)

    def backward_solve_dev(self, d_Z: int, ldz: int, nrhs: int, d_X: int, ldx: int, gather: bool = True) -> None:
        """X = P' L^-T Z (`F.UP \\ z`, src/workspace/backend.jl:281-284: the sampling path) with the factor sharded over the
        ranks; Z (full, column-major n x nrhs, in ELIMINATION order as CHOLMOD takes it) on every rank, X on rank 0 (gather = False:
        distributed, see solve_dev)."""
        for j0 in range(0, int(nrhs), 64):
            self._solve_pass(d_Z + 8 * j0 * ldz, ldz, min(64, nrhs - j0), d_X + 8 * j0 * ldx, ldx, backward_only=True, gather=gather)

    def valid_rows(self) -> np.ndarray:
        """Boolean mask over the n rows of X in the CALLER's (original) ordering: the rows that are final on THIS rank after a
        solve with gather = False -- the columns of its own subtrees and of every top front. The masks of all ranks cover every row;
        the top fronts' rows are valid everywhere."""
        n = self.be.n
        elim = np.zeros(n, bool)
        for o, r0, nr in self._sub_blocks:
            if o == self.rank:
                elim[r0:r0 + nr] = True
        for blocks in self._top_blocks:
            for _, r0, nr in blocks:
                elim[r0:r0 + nr] = True
        perm = np.asarray(self.be.ordering_permutation())      # perm[k] = original index of elimination position k
        out = np.zeros(n, bool)
        out[perm] = elim
        return out

    def needed_rows(self) -> np.ndarray:
        """Boolean mask over the n rows of B (solve_dev: the CALLER's ordering; backward_solve_dev takes Z in elimination order:
        use `needed_rows(elimination=True)`): the rows THIS rank reads -- the columns of its own subtrees and of the top fronts it
        OWNS. Nothing else of B is ever read here: a caller may leave every other row unset (B row-sharded over the ranks, never
        replicated; the masks of all ranks partition the rows). Tested with NaN in the other rows (one-GPU rehearsals)."""
        return self._row_mask(owned_top_only=True, elimination=False)

    def needed_rows_elimination(self) -> np.ndarray:
        return self._row_mask(owned_top_only=True, elimination=True)

    def _row_mask(self, owned_top_only: bool, elimination: bool) -> np.ndarray:
        n = self.be.n
        elim = np.zeros(n, bool)
        for o, r0, nr in self._sub_blocks:
            if o == self.rank:
                elim[r0:r0 + nr] = True
        for blocks in self._top_blocks:
            for o, r0, nr in blocks:
                if not owned_top_only or o == self.rank:
                    elim[r0:r0 + nr] = True
        if elimination:
            return elim
        perm = np.asarray(self.be.ordering_permutation())
        out = np.zeros(n, bool)
        out[perm] = elim
        return out

    def _solve_pass(self, d_B: int, ldb: int, nrhs: int, d_X: int, ldx: int, backward_only: bool, gather: bool = True) -> None:
        be = self.be
        if backward_only:
            be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 10)                     # z as is
        else:
            be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 0)                      # transpose in + own forward
            for k in range(self.K):
                self._p2p([(src, dst, 3, r0 * nrhs, nr * nrhs) for src, dst, r0, nr in self._w_items[k]], key=("w", k, nrhs))
                be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 100 + k)            # forward, top level k
        for k in range(self.K - 1, -1, -1):
            be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, (300 if backward_only else 200) + k)    # backward, top level k
            self._bcast_rows(self._top_blocks[k], nrhs)                          # x of that level's fronts -> everybody
        be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 12 if backward_only else 2)  # own backward
        if not gather:
            be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 3)                      # transpose out what this rank holds (valid_rows())
            return
        self._p2p([(o, 0, 2, r0 * nrhs, nr * nrhs) for o, r0, nr in self._sub_blocks], key=("home", nrhs))     # x of the owned subtrees -> rank 0
        if self.rank == 0:
            be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 3)                      # transpose out

    # ---- selected inversion -------------------------------------------------------------------------
    def selinv_compute(self) -> None:
        """Takahashi recursion, top-down over the sharded tree: a front needs the trailing block of its PARENT's inverse
        front; for an owner-crossing edge the parent's owner gathers that block (it holds both sources) into the
        shared arena layout and sends it to the child's owner. Levels without such edges run as one range."""
        be, e = self.be, self.edges
        nl = self.L0 + self.K
        be.selinv_phase(0)
        cut = sorted({int(l) for l in e["child_level"]}, reverse=True)     # levels that receive blocks
        hi = nl
        for l in cut:
            if hi > l + 1:
                be.selinv_phase(2, hi, l + 1)                    # my fronts of the levels above l
            be.selinv_phase(1, l)                                # gather for the other ranks' fronts of level l
            idx = np.flatnonzero(e["child_level"] == l)
            self._p2p([(int(e["dst"][i]), int(e["src"][i]), 0, int(e["zb_offset"][i]), int(e["cb_count"][i])) for i in idx])
            be.selinv_phase(2, l + 1, l)
            hi = l
        if hi > 0:
            be.selinv_phase(2, hi, 0)
        be.selinv_phase(3)

    def selinv_diag(self):
        """diag(Q^-1) in original ordering on every rank: all-reduce (sum) of the ranks' parts."""
        t = self.torch
        part = t.from_numpy(self.be.get_selinv_diag().copy())
        if not self.host_staging:
            part = part.to(self.dev)
        self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM)
        return part.cpu().numpy()

    def logdet(self) -> float:
        """log det Q: all-reduce (sum) of the ranks' partial sums over their own pivots."""
        t = self.torch
        part = t.tensor([self.be.logdet_partial()], dtype=t.float64, device="cpu" if self.host_staging else self.dev)
        self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM)
        if self._info_pending:                      # the pivot report of refactorize_dev(check=False) rides with this host round trip
            self._reduce_info()
        return float(part.item())

    def close(self):
        self.be.close()


def plan_summary(be, level_ms=None) -> dict:
    """Who owns how much, and bounds on the speed-up of the sharded factorisation (host-side, works on symbolic_only handles).

    flop_bound_speedup: time ~ max_r (flops of rank r's subtrees) + sum over top levels of the heaviest rank's flops in that
    level -- what the plan could reach if every front ran at one common flop rate.

    level_ms (optional): measured HIP-event time of every tree level of the UNSHARDED run (MI355XBackend.level_times under
    GMRFX_LEVEL_MARK=1; [0] = the sweep tasks, [1 + l] = level l of the unsharded schedule), for the factorisation alone or
    summed with the sweeps. It turns the plan into TIME bounds: a rank's subtree fronts cost their flop share of their level's
    measured time (those levels hold thousands of fronts: throughput-bound, the share is fair); the top fronts are the
    latency-bound chains at the top of the tree (potrf64 -> trsm -> gemm per 64 columns) that sharding does not shorten:
      time_bound_speedup_latency : the top fronts of a level still cost their full measured time, one after the other;
      time_bound_speedup_share   : the top fronts too only cost the heaviest rank's flop share of their level.
    The truth lies between the two; the exchanges are not in either.

    DISTRIBUTED TOP FRONTS (Symbolic::dist_fronts: factored by their whole group, 256-column blocks dealt cyclically) enter with
    their flops divided by the size of the group on every member; their time additionally keeps what does not shrink: the
    diagonal chain (CHAIN_MS_PER_64 per 64-column step, measured at cfg 2) and the block-column broadcasts (the panel once per
    member at BCAST_GBS, a conservative per-link xGMI figure)."""
    owner, top = be.shard_owner(with_top=True)
    sy = be.symbolic()
    c = np.diff(sy.super_first).astype(np.float64)
    r = np.diff(sy.row_ptr).astype(np.float64)
    m = r - c
    fl = c ** 3 / 3 + c * c * m + c * m * m
    W = int(owner.max()) + 1
    local = [float(fl[(owner == k) & ~top].sum()) for k in range(W)]
    df = be.shard_dist_fronts()
    dset = {int(s_): i for i, s_ in enumerate(df["front"])}
    isd = np.zeros(len(c), bool)
    isd[list(dset)] = True
    gsz = {s_: len(df["group"][i]) for s_, i in dset.items()}

    def heaviest(level_of, lv, cost, dcost):
        """max over ranks of (own undistributed top fronts of level lv at `cost`) + (distributed fronts of its groups at `dcost`)"""
        sel = top & (level_of == lv)
        best = 0.0
        for k in range(W):
            t = float(cost[sel & ~isd & (owner == k)].sum())
            t += sum(dcost(s_) for s_, i in dset.items() if level_of[s_] == lv and k in df["group"][i])
            best = max(best, t)
        return best

    t_top = sum(heaviest(sy.level, lv, fl, lambda s_: float(fl[s_]) / gsz[s_]) for lv in np.unique(sy.level[top]))
    total = float(fl.sum())
    out = {"world": W, "top_fronts": int(top.sum()), "top_flops": float(fl[top].sum()), "local_flops": local,
           "top_critical_flops": t_top, "flop_bound_speedup": total / (max(local) + t_top),
           "distributed_fronts": [{"supernode": s_, "cols": int(c[s_]), "rows": int(r[s_]), "group": gsz[s_], "flops": float(fl[s_])}
                                  for s_ in sorted(dset)]}
    if level_ms is not None:
        # levels of the UNSHARDED schedule (a sharded handle re-levels its top fronts): depth below the root, counted down from
        # the tree's height (csrc/symbolic.cpp "Levels by DEPTH below the root")
        ns = len(c)
        par = np.asarray(sy.super_parent)
        hgt = np.zeros(ns, np.int64)
        for s in range(ns):                         # children have smaller ids than their parents (postorder)
            if par[s] >= 0:
                hgt[par[s]] = max(hgt[par[s]], hgt[s] + 1)
        H = int(hgt.max())
        h = np.zeros(ns, np.int64)
        for s in range(ns - 1, -1, -1):
            h[s] = H if par[s] < 0 else h[par[s]] - 1
        if os.environ.get("GMRFX_TOP_BY_DEPTH") == "0":
            h = hgt
        lm = np.asarray(level_ms, dtype=np.float64)
        nl = int(h.max()) + 1
        if lm.size < nl + 1:
            raise ValueError(f"level_ms has {lm.size} entries, the tree has {nl} levels (+ 1 for the sweep tasks)")
        t_level = lm[1:nl + 1].copy()
        t_level[0] += lm[0]                         # the sweep tasks are bottom subtrees: count them with level 0
        F = np.array([fl[h == l].sum() for l in range(nl)])
        per_front = t_level[h] * fl / np.maximum(F[h], 1e-300)
        t_local = max(float(per_front[(owner == k) & ~top].sum()) for k in range(W))
        top_levels = np.unique(h[top])
        CHAIN_MS_PER_64, BCAST_GBS = 0.03, 100.0
        fixed = lambda s_: CHAIN_MS_PER_64 * c[s_] / 64.0 + 8.0 * (r[s_] * c[s_] - c[s_] ** 2 / 2.0) / (BCAST_GBS * 1e6)
        d_lat = lambda s_: min(float(per_front[s_]), float(per_front[s_]) / gsz[s_] + fixed(s_))
        # latency form: the undistributed top fronts of a level cost their full time one after the other (no concurrency between
        # the groups assumed), its distributed fronts the heaviest member's part
        t_latency = float(sum(per_front[top & ~isd & (h == l)].sum() + heaviest(h, l, 0.0 * per_front, d_lat) for l in top_levels))
        t_share = float(sum(heaviest(h, l, per_front, lambda s_: float(per_front[s_]) / gsz[s_]) for l in top_levels))
        # (the non-top fronts of a level that also holds top fronts are already in t_local with their share)
        t1 = float(t_level.sum())
        out.update({"measured_ms_one_gpu": t1, "time_bound_ms_latency": t_local + t_latency, "time_bound_ms_share": t_local + t_share,
                    "time_bound_speedup_latency": t1 / (t_local + t_latency), "time_bound_speedup_share": t1 / (t_local + t_share),
                    "top_levels": [int(l) for l in top_levels], "top_levels_ms": [float(t_level[l]) for l in top_levels]})
    return out
