#!/usr/bin/env python3
"""HIP-event time of every tree level of the factorisation and of the two sweeps on ONE GPU (GMRFX_LEVEL_MARK=1), for
cfg 2 (default) or cfg 4 (`cfg4 [N]`): the input of the TIME bound of a sharding plan (gmrfx/shard.py plan_summary,
tools/shard_bounds.py). Writes gpurun_out/level_ms_<cfg>.json.

    python3 tools/level_times.py [cfg2 [grid] | cfg4 [N]]"""
import json, os, sys
os.environ["GMRFX_LEVEL_MARK"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde

pipelined = "pipelined" in sys.argv      # the one-call step (gmrfx_refactorize_solve_dev): the levels of the factorisation with the forward sweep beside them
if pipelined:
    sys.argv.remove("pipelined")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if cfg == "cfg4":
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 126
    mesh = spde.grid_mesh_3d(N, N, N)
    Q = spde.matern_precision(mesh, 0, 0.4)
    name = f"cfg4_{N}cubed"
else:
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    mesh = spde.grid_mesh_2d(G, G, jitter=0.25, seed=0)
    Q = spde.matern_precision(mesh, 0, 0.2)
    name = f"cfg2_{G}"
n = Q.shape[0]
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
d_B = torch.randn((64, n), generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
d_X = torch.empty_like(d_B)
torch.cuda.synchronize()
runs = []
for k in range(3):
    if pipelined:
        be.refactorize_solve_dev(d_nz.data_ptr(), d_B.data_ptr(), n, 64, d_X.data_ptr(), n)
    else:
        be.refactorize_dev(d_nz.data_ptr())
        be.solve_dev(d_B.data_ptr(), n, 64, d_X.data_ptr(), n)
    s = be.stats()
    runs.append({"factor": be.level_times(0).tolist(), "fwd": be.level_times(1).tolist(), "bwd": be.level_times(2).tolist(),
                 "ms_factor": s["ms_factor"], "ms_solve": s["ms_solve"]})
    print(f"step {k}: factor {s['ms_factor']:.2f} ms (levels sum {sum(runs[-1]['factor']):.2f}), solve {s['ms_solve']:.2f} ms "
          f"(fwd levels {sum(runs[-1]['fwd']):.2f}, bwd levels {sum(runs[-1]['bwd']):.2f})", flush=True)
out = {"workload": name, "n": int(n), "nrhs": 64, "note": "[0] = the sweep tasks, [1 + l] = tree level l (height above the leaves); the last of 3 steps",
       **runs[-1]}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"level_ms_{name}{'_pipelined' if pipelined else ''}.json"), "w"))
print(json.dumps(out))
