"""TEST HELPER (moved out of the package in round 3: the product drives replicas from bench.py itself). Replica sharding over
INDEPENDENT units -- hyper-parameter points / posterior workspaces, the reference's
`WorkspacePool` pattern (src/workspace/workspace_pool.jl:42-119) -- so ranks exchange no data
on the data path; the only collectives are the barrier / MAX-reduce of the timing and an
all-gather of the per-unit scalars (logdet / logpdf) at the end."""
from __future__ import annotations

import time
from typing import Callable, List, Sequence


def shard_units(n_units: int, rank: int, world: int) -> range:
    """Contiguous block partition, sizes differ by at most one."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def run_sharded(units: Sequence, work: Callable, dist=None, sync: Callable = lambda: None):
    """Each rank runs `work(unit)` on its shard; returns (results for ALL units in order,
    max-over-ranks wall time). `dist` is torch.distributed (already initialised) or None."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    mine = shard_units(len(units), rank, world)
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    local = [work(units[i]) for i in mine]
    sync()
    elapsed = time.perf_counter() - t0
    if dist is None:
        return local, elapsed
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64)
    backend = dist.get_backend()
    if backend == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered: List = [None] * world
    dist.all_gather_object(gathered, local)
    return [x for part in gathered for x in part], float(t.item())
