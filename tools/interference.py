#!/usr/bin/env python3
"""What slows the factorisation when other work runs beside it on the same GPU? The factorisation of cfg 2 is timed (HIP events of
the library) alone and with a background thread that keeps a second stream busy with one KIND of work:
  tiny    an endless sequence of empty-ish kernels (1 workgroup each): dispatch / command-processor interference only
  alu     large FP64 matrix products (hipBLASLt through torch): compute units + L2
  stream  large device-to-device copies: HBM bandwidth
  small   many medium elementwise kernels over 64 MB: a mix, like the level kernels of a sweep
    python3 tools/interference.py [grid=1000] [steps=30]"""
import os, sys, threading, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np, torch
import gmrfx
from gmrfx import spde

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mesh = spde.grid_mesh_2d(G, G, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, 0, 0.2)
n = Q.shape[0]
dev = torch.device("cuda", 0)
be = gmrfx.MI355XBackend(Q, coords=mesh.points, factorize=False)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to(dev)
side = torch.cuda.Stream(device=dev, priority=0)          # torch: 0 = default (lowest) priority
A = torch.randn((4096, 4096), dtype=torch.float64, device=dev)
B = torch.randn((4096, 4096), dtype=torch.float64, device=dev)
big = torch.empty(64 * 1024 * 1024 // 8, dtype=torch.float64, device=dev)       # 64 MB
big2 = torch.empty_like(big)
huge = torch.empty(1024 * 1024 * 1024 // 8, dtype=torch.float64, device=dev)    # 1 GB
huge2 = torch.empty_like(huge)
one = torch.zeros(64, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
stop = threading.Event()


def background(kind):
    with torch.cuda.stream(side):
        while not stop.is_set():
            if kind == "tiny":
                for _ in range(64):
                    one.add_(1.0)
            elif kind == "alu":
                torch.matmul(A, B)
            elif kind == "stream":
                huge2.copy_(huge)
            elif kind == "small":
                for _ in range(16):
                    big2.copy_(big); big.add_(1.0)
            side.synchronize()


def measure():
    ts = []
    for _ in range(steps):
        be.refactorize_dev(d_nz.data_ptr())
        ts.append(be.stats()["ms_factor"])
    return float(np.median(ts))


out = {"workload": f"{G}x{G} mesh", "alone_ms": None}
for _ in range(3):
    be.refactorize_dev(d_nz.data_ptr())
out["alone_ms"] = measure()
print("alone", round(out["alone_ms"], 3), flush=True)
for kind in ("tiny", "alu", "stream", "small"):
    stop.clear()
    th = threading.Thread(target=background, args=(kind,))
    th.start()
    time.sleep(0.2)
    out[kind + "_ms"] = measure()
    stop.set(); th.join()
    torch.cuda.synchronize()
    print(kind, round(out[kind + "_ms"], 3), flush=True)
out["alone_again_ms"] = measure()
print(json.dumps(out))
