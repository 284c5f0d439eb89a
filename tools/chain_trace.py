import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
os.environ["GMRFX_CHAIN_TRACE"] = "1"
import numpy as np, torch, gmrfx
from gmrfx import spde
mesh = spde.grid_mesh_2d(1000, 1000, jitter=0.25, seed=0)
Q = spde.matern_precision(mesh, smoothness=0, range_=0.2)
be = gmrfx.MI355XBackend(Q, coords=mesh.points, device=0, factorize=False)
d_nz = torch.from_numpy(np.ascontiguousarray(Q.data)).to("cuda:0")
for _ in range(3):
    sys.stderr.write("---- refactorize\n")
    be.refactorize_dev(d_nz.data_ptr())
