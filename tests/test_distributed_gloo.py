"""CPU, world_size 2, gloo: the N>1 path of bench.py -- units (independent hyper-parameter
points) are sharded over ranks with no data-path collective; timing is MAX-reduced and the
per-unit scalars all-gathered. Numeric work per unit is done here by the test's host walk
(tests/mf_hostsim.py) because this container has no GPU; on the GPU box the same driver
function runs MI355XBackend per rank (bench.py)."""
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
    from replica_sharding import run_sharded, shard_units
    from mf_hostsim import HostSim
    mesh = spde.grid_mesh_2d(12, 12, jitter=0.2)
    Q0 = spde.matern_precision(mesh, 0, 0.4)
    be = gmrfx.MI355XBackend(Q0, coords=mesh.points, symbolic_only=True)   # symbolic once per rank
    sy = be.symbolic()
    taus = [0.5, 1.0, 2.0, 3.0, 4.5]

    def work(tau):   # one "unit": logdet of tau*Q on the shared symbolic structure
        return HostSim(sy, Q0.shape[0], np.asarray(Q0.data) * tau).factor().logdet()

    res, t = run_sharded(taus, work, dist)
    if rank == 0:
        q.put((res, t, [list(shard_units(5, r, world)) for r in range(world)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_replicas():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res, t, shards = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert shards == [[0, 1, 2], [3, 4]]
    sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
    from gmrfx import spde
    mesh = spde.grid_mesh_2d(12, 12, jitter=0.2)
    Q0 = spde.matern_precision(mesh, 0, 0.4)
    ld0 = np.linalg.slogdet(Q0.toarray())[1]
    want = [ld0 + 144 * np.log(tau) for tau in [0.5, 1.0, 2.0, 3.0, 4.5]]
    assert np.allclose(res, want, rtol=1e-10)
    assert t > 0


def test_shard_units_partition_properties():
    sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
    from replica_sharding import shard_units
    for n in (0, 1, 7, 8, 9, 64):
        for w in (1, 2, 3, 8):
            parts = [list(shard_units(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
