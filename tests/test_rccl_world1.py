"""First execution of the RCCL device path of the sharded driver (gmrfx/shard.py), on the ONE GPU the test pool gives.

The sharded protocol had only ever run through its gloo / host-staging twin (tests/test_shard_gloo.py, the one-GPU
rehearsals of tests/test_gpu_parity.py): RCCL refuses two ranks on one device. A ONE-rank "nccl" group is runnable here, and a
sharded handle of one rank with a forced top (gmrfx_opts.shard_min_top) gives it the whole phase sequence to drive. The work is
done by tests/rccl_world1_child.py in a FRESH process (never a re-exec of a process that has touched the GPU); this file holds the
CPU-side checks of the one-rank plan and the GPU test that launches the child."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gmrfx
from gmrfx import spde

HERE = os.path.dirname(os.path.abspath(__file__))


def test_one_rank_plan_with_a_forced_top_is_a_sharded_plan_without_exchanges():
    """symbolic only (CPU): shard_world = 1 + shard_min_top > 0 -> top fronts and top levels exist, everything is owned by rank 0,
    no cross edges, no transfers, no distributed fronts; shard_min_top = 0 stays the plain unsharded handle"""
    mesh = spde.grid_mesh_2d(60, 50, jitter=0.2, seed=0)
    Q = spde.matern_precision(mesh, 0, 0.3)
    plain = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True)
    assert plain.shard_info()["n_top_fronts"] == 0 and plain.shard_info()["n_top_levels"] == 0
    be = gmrfx.MI355XBackend(Q, coords=mesh.points, symbolic_only=True, shard_rank=0, shard_world=1, shard_min_top=3)
    info = be.shard_info()
    assert info["n_top_fronts"] >= 3 and info["n_top_levels"] >= 1 and info["n_edges"] == 0
    owner, top = be.shard_owner(with_top=True)
    assert set(owner.tolist()) == {0} and int(top.sum()) == info["n_top_fronts"]
    assert be.shard_transfers()["src"].size == 0 and len(be.shard_dist_fronts()["front"]) == 0
    o2, r0, nr, lv = be.shard_rows(2)
    assert len(o2) == info["n_top_fronts"] and set(o2.tolist()) == {0}
    o3, s0, sn, _ = be.shard_rows(3)
    # the subtrees' and the top fronts' columns tile 0 .. n - 1
    cov = np.zeros(Q.shape[0], int)
    for a, k in list(zip(r0, nr)) + list(zip(s0, sn)):
        cov[a:a + k] += 1
    assert (cov == 1).all()
    # same elimination order, same fill as the unsharded analysis
    assert np.array_equal(be.ordering_permutation(), plain.ordering_permutation())
    assert be.stats()["nnz_l"] == plain.stats()["nnz_l"]


@pytest.mark.gpu
def test_rccl_path_runs_on_a_one_rank_group():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29931", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_child.py")], capture_output=True, text=True, timeout=420, env=env)
    line = next((l for l in r.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")), None)
    assert line is not None, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads(line[len("RCCL_WORLD1 "):])
    assert res.get("ok"), res.get("error", res)
    assert res["backend"] == "nccl" and res["rccl_ranks"] == 1 and res["top_levels"] >= 1 and res["top_fronts"] >= 3
    assert res["factor_bit_identical"], res
    assert res["solve_bit_identical"], res
    assert res["backward_bit_identical"], res
    assert abs(res["logdet"] - res["logdet_ref"]) <= 1e-12 * abs(res["logdet_ref"])
    assert res["residual"] < 1e-10 and res["selinv_diag_maxrel"] < 1e-10
    assert all(res["collectives_on_views_identity"].values()), res
    assert res["self_p2p_bytes_equal"], res


PKG = os.path.join(os.path.dirname(HERE), "gaussianmarkovrandomfields.jl_amd")


def test_native_rccl_driver_library_exports_its_header():
    """libgmrfx_rccl.so (csrc/rccl_driver.cpp: the sharded protocol over RCCL without Python, include/gmrfx_rccl.h) is built by
    __graft_entry__.build() and exports every symbol its header declares; it links libgmrfx.so through the public C ABI only."""
    import ctypes
    import re
    lib = os.path.join(PKG, "libgmrfx_rccl.so")
    assert os.path.exists(lib), "libgmrfx_rccl.so has not been built (make -C gaussianmarkovrandomfields.jl_amd rccl)"
    hdr = open(os.path.join(os.path.dirname(HERE), "include", "gmrfx_rccl.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    syms = sorted(set(re.findall(r"\b(gmrfx_rccl_[a-z_0-9]+)\s*\(", hdr)))
    assert len(syms) >= 9
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    for s in syms:
        assert re.search(r"\bT %s\b" % s, out), f"{s} declared in include/gmrfx_rccl.h but not exported"
    # the driver sits ABOVE the C ABI: it may not reach into the library's internals
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    internal = [l for l in und.splitlines() if "gmrfx" in l and not re.search(r"\bU gmrfx_[a-z_0-9]+$", l.strip())]
    assert not internal, internal


@pytest.mark.gpu
def test_native_rccl_driver_on_a_one_rank_communicator():
    """tools/rccl_driver_test.cpp: no Python in the process -- a sharded handle of one rank through gmrfx_rccl_refactorize / _solve /
    _backward_solve / _logdet / _selinv_diag on a one-rank RCCL communicator against an unsharded handle (bit for bit / 1e-12)."""
    exe = os.path.join(PKG, "rccl_driver_test")
    assert os.path.exists(exe), "rccl_driver_test has not been built (make -C gaussianmarkovrandomfields.jl_amd rccl)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "rccl_driver_test: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
