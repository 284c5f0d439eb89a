import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianmarkovrandomfields.jl_amd"))
import numpy as np
import gmrfx
from gmrfx import spde
mesh = spde.grid_mesh_2d(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 140, jitter=0.25, seed=5)
Q = spde.matern_precision(mesh, 0, 0.2)
os.environ["GMRFX_CHAIN_MAX_FRONTS"] = "0"
a = gmrfx.MI355XBackend(Q, coords=mesh.points)
os.environ["GMRFX_CHAIN_MAX_FRONTS"] = "32"
b = gmrfx.MI355XBackend(Q, coords=mesh.points)
fa, fb = a.factor_values(), b.factor_values()
sy = a.symbolic()
print("info", a.last_info, b.last_info, "len", len(fa))
bad = np.flatnonzero(~((fa == fb) | (np.isnan(fa) & np.isnan(fb))))
print("differing entries:", len(bad))
pp = np.asarray(sy.panel_ptr)
sn = np.searchsorted(pp, bad, side="right") - 1
lv = np.asarray(sy.level)
first_level = lv[np.unique(sn)].min() if len(bad) else -1
for s in np.unique(sn):
    if lv[s] != first_level: continue
    c = sy.super_first[s + 1] - sy.super_first[s]; r = sy.row_ptr[s + 1] - sy.row_ptr[s]
    ld = (pp[s + 1] - pp[s]) // c
    off = bad[sn == s] - pp[s]
    cols, rows = off // ld, off % ld
    k = np.lexsort((rows, cols))[0]
    print(f"supernode {s}: c={c} r={r} ld={ld} level={lv[s]} first bad (row,col)=({rows[k]},{cols[k]}), bad cols {cols.min()}..{cols.max()}, rows {rows.min()}..{rows.max()}, count {len(off)}; a={fa[bad[sn==s][k]]:.6e} b={fb[bad[sn==s][k]]:.6e}")
    # per 64x64 tile summary
    tiles = sorted(set(zip((rows // 64).tolist(), (cols // 64).tolist())))
    print("   bad tiles (row tile, col block):", tiles[:40])
