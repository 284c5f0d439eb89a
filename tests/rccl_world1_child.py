"""Child process of tests/test_rccl_world1.py: the RCCL device path of gmrfx/shard.py executed on ONE GPU.

A one-rank "nccl" process group (torch.distributed's nccl backend IS RCCL on ROCm) on cuda:0, a sharded handle of one rank with a
forced top (gmrfx_opts.shard_min_top), and
  (a) broadcast / all-reduce on the zero-copy views (`__cuda_array_interface__`) of the library's arena, X and W buffers,
  (b) a self-addressed batch_isend_irecv between two disjoint views of the arena,
  (c) refactorize_dev(check=False) -> solve_dev(gather=False) -> logdet() on torch's current stream with asynchronous phases,
      compared with an unsharded handle,
so that `_p2p`'s cached-P2POp branch, `_bcast_rows`' async broadcasts of in-place views, `_reduce_info` / `logdet` on device tensors
and `init_process_group("nccl", device_id=...)` have run at least once (rounds 1-5: only their gloo / host-staging twins had).
What stays unexercised: a transfer that really crosses ranks (two GPUs). Replaces nothing in the reference -- its solver has one
address space (src/workspace/backend.jl:165-209 is what the exchange stands in for).

Prints one JSON line. Run as a fresh process (it initialises the GPU and RCCL itself)."""
import json
import os
import sys
import traceback


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "gaussianmarkovrandomfields.jl_amd"), os.path.join(root, "oracle"), here):
        sys.path.insert(0, p)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29931")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    out = {"ok": False}
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out["backend"] = dist.get_backend()
    import gmrfx as g
    from gmrfx import spde, shard

    m = spde.grid_mesh_2d(120, 120, jitter=0.25, seed=2)
    Q = spde.matern_precision(m, 0, 0.2)
    n = Q.shape[0]
    sf = shard.ShardedFactor(Q, dist, device=0, coords=m.points, shard_min_top=3)
    assert not sf.host_staging and sf.world == 1 and sf.K >= 1, (sf.host_staging, sf.world, sf.K)
    out["top_levels"] = sf.K
    out["top_fronts"] = int(sf.info["n_top_fronts"])
    d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)

    # (c) the step bench.py --gpus N times, on torch's current stream, no host round trip before logdet()
    nrhs = 64
    Bh = torch.randn((nrhs, n), generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    d_B = Bh.to(dev)
    d_X = torch.full_like(d_B, float("nan"))
    for _ in range(2):
        assert sf.refactorize_dev(d_nz.data_ptr(), check=False) == 0 and sf._info_pending
        sf.solve_dev(d_B.data_ptr(), n, nrhs, d_X.data_ptr(), n, gather=False)
        ld = sf.logdet()
        assert sf.last_info == 0 and not sf._info_pending
    torch.cuda.synchronize()
    assert bool(sf.valid_rows().all())                       # one rank: every row is its own
    ref = g.MI355XBackend(Q, coords=m.points, device=0)
    d_Xr = torch.zeros_like(d_B)
    torch.cuda.synchronize()
    ref.solve_dev(d_B.data_ptr(), n, nrhs, d_Xr.data_ptr(), n)
    torch.cuda.synchronize()
    out["solve_bit_identical"] = bool(torch.equal(d_X, d_Xr))
    out["solve_maxdiff"] = float((d_X - d_Xr).abs().max())
    out["logdet"] = ld
    out["logdet_ref"] = ref.compute_logdet()
    # the factor, panel by panel (a sharded handle has its own panel offsets)
    sy, rsy = sf.be.symbolic(), ref.symbolic()
    vals, rv = sf.be.factor_values(), ref.factor_values()
    same = True
    for s in range(len(sy.panel_ld)):
        a, ar = int(sy.panel_ptr[s]), int(rsy.panel_ptr[s])
        c, r, ldp = int(sy.super_first[s + 1] - sy.super_first[s]), int(sy.row_ptr[s + 1] - sy.row_ptr[s]), int(sy.panel_ld[s])
        Pa = vals[a:a + ldp * c].reshape(c, ldp).T[:r]
        Pb = rv[ar:ar + ldp * c].reshape(c, ldp).T[:r]
        if not np.array_equal(np.tril(Pa), np.tril(Pb)):
            same = False
            break
    out["factor_bit_identical"] = same
    X = d_X.cpu().numpy().T
    out["residual"] = float(np.linalg.norm(Q @ X - Bh.numpy().T) / np.linalg.norm(Bh.numpy()))
    # backward-only solve (the sampling path) through phases 10 / 300 + k / 12
    d_Z = torch.full_like(d_B, float("nan"))
    sf.backward_solve_dev(d_B.data_ptr(), n, nrhs, d_Z.data_ptr(), n, gather=False)
    d_Zr = torch.zeros_like(d_B)
    torch.cuda.synchronize()
    ref.backward_solve_dev(d_B.data_ptr(), n, nrhs, d_Zr.data_ptr(), n)
    torch.cuda.synchronize()
    out["backward_bit_identical"] = bool(torch.equal(d_Z, d_Zr))
    out["backward_maxdiff"] = float((d_Z - d_Zr).abs().max())
    # sharded selected inversion: phases + the all-reduced diagonal on a device tensor
    sf.selinv_compute()
    sd = sf.selinv_diag()
    rd = ref.get_selinv_diag()
    out["selinv_diag_maxrel"] = float(np.abs(sd - rd).max() / np.abs(rd).max())

    # (a) collectives on the zero-copy views of library memory (contents are scratch from here on)
    arena_doubles = int(sf.be.stats()["bytes_cb_arena"]) // 8
    cnt = int(min(1 << 16, arena_doubles // 2))
    assert cnt >= 1024, arena_doubles
    views = {"arena": sf._view(0, 0, cnt), "X": sf._view(2, 0, min(cnt, n * 64)), "W": sf._view(3, 0, 4096)}
    coll = {}
    for name, v in views.items():
        base = sf.be.device_ptr({"arena": 0, "X": 2, "W": 3}[name])
        assert v.data_ptr() == base and v.is_cuda and v.dtype == torch.float64, name          # really the library's memory, no copy
        v.copy_(torch.arange(v.numel(), dtype=torch.float64, device=dev) * 0.5 + 1.0)
        before = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        w = dist.broadcast(v, src=0, async_op=True)
        w.wait()
        torch.cuda.synchronize()
        coll[name] = bool(torch.equal(v, before))
    out["collectives_on_views_identity"] = coll
    # (b) self-addressed batched send + recv between two disjoint views of the arena
    a, b = sf._view(0, 0, cnt), sf._view(0, cnt, cnt)
    a.copy_(torch.arange(cnt, dtype=torch.float64, device=dev) * 3.0 - 7.0)
    b.zero_()
    torch.cuda.synchronize()
    ops = [dist.P2POp(dist.irecv, b, 0, tag=0), dist.P2POp(dist.isend, a, 0, tag=0)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    torch.cuda.synchronize()
    out["self_p2p_bytes_equal"] = bool(torch.equal(a, b))
    # and through the driver's own cached-P2POp branch: a recurring list whose single item this rank both... is skipped (src == dst):
    # the branch must build an empty list and return without posting
    sf._p2p([(0, 0, 0, 0, cnt)], key=("selftest", 0))
    out["p2p_cache_keys"] = len(sf._p2p_cache)
    t = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    out["rccl_ranks"] = int(dist.get_world_size())
    out["ok"] = True
    sf.close()
    ref.close()
    dist.destroy_process_group()
    return out


if __name__ == "__main__":
    try:
        res = main()
    except Exception:
        res = {"ok": False, "error": traceback.format_exc()[-3000:]}
    print("RCCL_WORLD1 " + json.dumps(res), flush=True)
    sys.exit(0 if res.get("ok") else 1)
