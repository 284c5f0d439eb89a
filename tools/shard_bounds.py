#!/usr/bin/env python3
"""Flop and TIME bounds of the sharding plan at 2 / 4 / 8 ranks for cfg 2 and cfg 4, from the per-level times measured on one
GPU (tools/level_times.py -> profiles/r06_level_ms_<cfg>.json; GMRFX_PROFILE_ROUND picks another round's files). Host only (symbolic-only handles).

    python3 tools/shard_bounds.py [cfg2|cfg4] ...   -> prints a table and one JSON object per configuration"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np
import gmrfx
from gmrfx import spde, shard


def run(cfg):
    if cfg == "cfg4":
        mesh = spde.grid_mesh_3d(126, 126, 126); Q = spde.matern_precision(mesh, 0, 0.4); name = "cfg4_126cubed"
    else:
        mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0); Q = spde.matern_precision(mesh, 0, 0.2); name = "cfg2_1000"
    lv = json.load(open(os.path.join(ROOT, "profiles", f"{os.environ.get('GMRFX_PROFILE_ROUND', 'r06')}_level_ms_{name}.json")))
    res = {"workload": name, "one_gpu_ms": {"factor": sum(lv["factor"]), "fwd": sum(lv["fwd"]), "bwd": sum(lv["bwd"])}, "ranks": {}}
    step = np.asarray(lv["factor"]) + np.asarray(lv["fwd"]) + np.asarray(lv["bwd"])
    print(f"== {name}: one GPU factor {sum(lv['factor']):.2f} ms + sweeps {sum(lv['fwd']) + sum(lv['bwd']):.2f} ms")
    print(" ranks  top fronts  flop bound   factor: time bound (latency .. share)   step: time bound (latency .. share)")
    for W in (2, 4, 8):
        be = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True, shard_rank=0, shard_world=W)
        pf = shard.plan_summary(be, lv["factor"])
        ps = shard.plan_summary(be, step)
        be.close()
        res["ranks"][W] = {"top_fronts": pf["top_fronts"], "flop_bound_speedup": pf["flop_bound_speedup"],
                           "factor": {k: pf[k] for k in ("time_bound_speedup_latency", "time_bound_speedup_share", "time_bound_ms_latency", "time_bound_ms_share")},
                           "step": {k: ps[k] for k in ("time_bound_speedup_latency", "time_bound_speedup_share", "time_bound_ms_latency", "time_bound_ms_share")},
                           "top_levels": pf["top_levels"], "top_levels_ms_factor": pf["top_levels_ms"]}
        print(f" {W:5d} {pf['top_fronts']:11d} {pf['flop_bound_speedup']:11.2f}   {pf['time_bound_speedup_latency']:8.2f} .. {pf['time_bound_speedup_share']:5.2f}"
              f"                       {ps['time_bound_speedup_latency']:8.2f} .. {ps['time_bound_speedup_share']:5.2f}")
    print(json.dumps(res))


def memory(cfg, W=8):
    """per-rank HBM of the sharded plan from symbolic-only handles (one analysis per rank): factor panels, contribution-block arena
    (own blocks + the exchange region), right-hand-side panels X / X2 (n x 64 each)"""
    if cfg == "cfg4":
        mesh = spde.grid_mesh_3d(126, 126, 126); Q = spde.matern_precision(mesh, 0, 0.4); name = "cfg4_126cubed"
    else:
        mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0); Q = spde.matern_precision(mesh, 0, 0.2); name = "cfg2_1000"
    n = Q.shape[0]
    one = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True)
    s1 = one.stats(); perm = one.ordering_permutation(); one.close()
    print(f"== {name}: per-rank memory at world {W} (GB): factor panels, contribution-block arena, X + X2; unsharded: "
          f"{s1['bytes_factor'] / 1e9:.1f} + {s1['bytes_cb_arena'] / 1e9:.1f} + {2 * n * 64 * 8 / 1e9:.1f}", flush=True)
    rows = []
    for r in range(W):
        be = gmrfx.MI355XBackend(Q, ordering=perm, symbolic_only=True, shard_rank=r, shard_world=W)
        st = be.stats(); be.close()
        rows.append((st["bytes_factor"] / 1e9, st["bytes_cb_arena"] / 1e9, 2 * n * 64 * 8 / 1e9))
        print(f"  rank {r}: {rows[-1][0]:7.2f} + {rows[-1][1]:7.2f} + {rows[-1][2]:5.2f} = {sum(rows[-1]):7.2f} GB", flush=True)
    print(f"  largest rank: {max(sum(x) for x in rows):.2f} GB; sum of the factor shares {sum(x[0] for x in rows):.2f} GB")


args = sys.argv[1:] or ["cfg2"]
if args[0] == "memory":
    for c in (args[1:] or ["cfg2"]):
        memory(c)
else:
    for c in args:
        run(c)
