"""One factorisation sharded over the ranks of a torch.distributed process group (one GPU each).

SURVEY section 8(e): the supernodal tree is cut top-down into subtrees that are dealt to the ranks
(`Symbolic::owner`, csrc/symbolic.cpp); every rank factors its subtrees, the contribution blocks of
the subtree roots -- the Schur complements the fronts above them assemble -- travel to rank 0 over the
process group (RCCL point-to-point on MI355X nodes, gloo in the CPU-side rehearsal), rank 0 factors the
top fronts, and log det Q is an all-reduce of the ranks' partial sums. This is the exchange step the
reference's multifrontal solvers (CHOLMOD behind src/workspace/backend.jl:165-189) do inside one
address space.

Device buffers are handed to torch.distributed without copies: the contribution-block arena of the
library is wrapped as a torch tensor through __cuda_array_interface__.
"""
from __future__ import annotations

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
        self.owner, self.off, self.cnt = self.be.shard_cb_blocks()
        self.host_staging = dist.get_backend() == "gloo"      # rehearsal: gloo moves host tensors only
        self._rows = None

    def _cb_view(self, off: int, cnt: int):
        base = self.be.device_ptr(0)
        return self.torch.as_tensor(_DevView(base + 8 * int(off), int(cnt)), device=self.dev)

    def _exchange_cb(self):
        """Subtree-root contribution blocks -> rank 0, written in place into its arena."""
        t, dist = self.torch, self.dist
        reqs = []
        for k in range(len(self.owner)):
            src = int(self.owner[k])
            if src == 0:
                continue
            if self.rank == src:
                v = self._cb_view(self.off[k], self.cnt[k])
                dist.send(v.cpu() if self.host_staging else v, dst=0, tag=k)
            elif self.rank == 0:
                v = self._cb_view(self.off[k], self.cnt[k])
                if self.host_staging:
                    buf = t.empty(int(self.cnt[k]), dtype=t.float64)
                    dist.recv(buf, src=src, tag=k)
                    v.copy_(buf)
                else:
                    dist.recv(v, src=src, tag=k)
        t.cuda.synchronize(self.dev)
        return reqs

    def refactorize_dev(self, d_nzval_ptr: int) -> None:
        self.be.refactorize_phase_dev(d_nzval_ptr, 0)
        self._exchange_cb()
        self.be.refactorize_phase_dev(d_nzval_ptr, 1)

    # ---- sharded solve ---------------------------------------------------------------------------
    def _rows_view(self, which: int, row0: int, nrows: int, nrhs: int):
        base = self.be.device_ptr(which)
        return self.torch.as_tensor(_DevView(base + 8 * int(row0) * nrhs, int(nrows) * nrhs), device=self.dev)

    def _move_rows(self, which: int, blocks, nrhs: int, to_root: bool):
        """to_root: every owner sends its row blocks to rank 0 (in place, same rows of rank 0's buffer);
        otherwise rank 0 broadcasts its row blocks to everybody."""
        t, dist = self.torch, self.dist
        owner, r0, nr = blocks
        for k in range(len(owner)):
            if to_root:
                src = int(owner[k])
                if src == 0 or self.rank not in (0, src):
                    continue
                v = self._rows_view(which, r0[k], nr[k], nrhs)
                if self.rank == src:
                    dist.send(v.cpu() if self.host_staging else v, dst=0, tag=k)
                elif self.host_staging:
                    buf = t.empty(v.numel(), dtype=t.float64)
                    dist.recv(buf, src=src, tag=k)
                    v.copy_(buf)
                else:
                    dist.recv(v, src=src, tag=k)
            else:
                v = self._rows_view(which, r0[k], nr[k], nrhs)
                if self.host_staging:
                    buf = v.cpu() if self.rank == 0 else t.empty(v.numel(), dtype=t.float64)
                    dist.broadcast(buf, src=0)
                    if self.rank != 0:
                        v.copy_(buf)
                else:
                    dist.broadcast(v, src=0)
        t.cuda.synchronize(self.dev)

    def solve_dev(self, d_B: int, ldb: int, nrhs: int, d_X: int, ldx: int) -> None:
        """Q X = B with the factor sharded over the ranks; B (full, column-major n x nrhs) on every rank,
        X (full) is produced on rank 0. 1..64 right-hand sides per call."""
        be = self.be
        if self._rows is None:
            self._rows = {k: be.shard_rows(k) for k in (1, 2, 3)}
        be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 0)          # transpose in + own forward
        self._move_rows(3, self._rows[1], nrhs, True)            # W of the subtree roots -> rank 0
        be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 1)          # the top: forward, backward (rank 0)
        self._move_rows(2, self._rows[2], nrhs, False)           # x of the top fronts -> everybody
        be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 2)          # own backward
        self._move_rows(2, self._rows[3], nrhs, True)            # x of the owned subtrees -> rank 0
        if self.rank == 0:
            be.solve_phase_dev(d_B, ldb, nrhs, d_X, ldx, 3)      # transpose out

    def logdet(self) -> float:
        """log det Q: all-reduce (sum) of the ranks' partial sums over their own pivots."""
        t = self.torch
        part = t.tensor([self.be.logdet_partial()], dtype=t.float64, device="cpu" if self.host_staging else self.dev)
        self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM)
        return float(part.item())

    def close(self):
        self.be.close()


def plan_summary(be) -> dict:
    """Who owns how much (host-side, works on symbolic_only handles too)."""
    owner = be.shard_owner()
    sy = be.symbolic()
    c = np.diff(sy.super_first).astype(np.float64)
    r = np.diff(sy.row_ptr).astype(np.float64)
    m = r - c
    fl = c ** 3 / 3 + c * c * m + c * m * m
    out = {"top_fronts": int((owner == -1).sum()), "top_flops": float(fl[owner == -1].sum())}
    for k in range(int(owner.max()) + 1):
        out[f"rank{k}_flops"] = float(fl[owner == k].sum())
    return out
