"""The bench.py contract on a GPU box: ONE JSON line with the headline metric, the roofline of the dominant kernel (live HIP
events), the secondary roofline objects, the phases and the correctness evidence -- on a reduced grid so that it runs in
seconds (the driver runs the default 1000 x 1000 configuration). Also the two-rank rehearsal (two processes on this one GPU,
gloo): the N > 1 line carries the sharded strong-scaling result as the headline and the replicas next to it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, timeout=600, expect_rc=0, extra_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == expect_rc, (p.returncode, p.stderr[-2000:])
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run(["--grid", "300", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_factor", "roofline_sweep", "phases_ms", "check"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["unit"] == "DoF/s"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["n"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6 and r["kernel"] == "k_syrk_cb_rec"
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches_per_step"] >= 1 and r["ms_per_step"] > 0.0          # live HIP events around every launch
    assert d["roofline_sweep"]["bound"] == "hbm" and d["roofline_sweep"]["peak"] == 8000.0 and 0.0 < d["roofline_sweep"]["frac"] < 1.0
    assert 0.0 < d["roofline_factor"]["frac"] < 1.0
    # round 6: the sweep's bytes use the TRUE nnz(L); a ceiling that concedes the multifrontal schedule is printed beside the fraction
    rs = d["roofline_sweep"]
    assert rs["frac_on_stored_entries"] >= rs["frac"] and rs["copy_peak_measured"] > 500.0
    mb = rs["model_bound_ms"]
    assert 0.0 < mb["fwd"] and 0.0 < mb["bwd"] and abs(mb["mean"] - 0.5 * (mb["fwd"] + mb["bwd"])) < 1e-12
    assert 0.0 < rs["frac_of_model_bound"] < 1.5 and rs["model_bound"]["fwd_launches"] >= 1 and rs["model_bound"]["handoff_bytes_fwd"] >= 0.0
    assert rs["model_bound_ms_at_guide_copy_peak"]["copy_peak_gbs"] == 6290.0
    assert d["check"]["rel_residual"] < 1e-10 and d["check"]["fail_col"] == -1
    ph = d["phases_ms"]
    assert ph["factor"] > 0 and ph["solve"] > 0 and abs(ph["solve"] - (ph["solve_fwd"] + ph["solve_bwd"] + ph["solve_perm"])) < 0.3 * ph["solve"]
    assert d["logpdf_ms"] > 0 and d["logpdf_relerr_vs_host"] < 1e-12
    # the timed step is the one pipelined call; the phases one after the other are measured beside it
    assert "gmrfx_refactorize_solve_dev" in d["step_call"] and d["ms_per_step_separate_calls"] > 0
    pp = d["pipelined_phases_ms"]
    assert pp["factor"] > 0 and pp["behind_factor"] > 0 and pp["forward_left_behind_factor"] <= ph["solve_fwd"] + 0.05
    assert d["roofline"]["achieved_alone"] > 0


def test_two_rank_rehearsal_line_is_sharded_strong_scaling():
    d = _run(["--gpus", "2", "--rehearse", "--grid", "200", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-logpdf", "--no-cfg3",
              "--cfg4-grid", "14"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "replicas" in d and d["replicas"]["scaling"] == "weak"
    sh = d["sharded"]
    assert "error" not in sh and sh["check"]["info"] == 0 and sh["check"]["rel_residual"] < 1e-10
    assert abs(sh["check"]["logdet"] - d["check"]["logdet"]) < 1e-11 * abs(d["check"]["logdet"])
    assert d["value"] == sh["value"] and d["ms_per_step"] == sh["ms_per_step"]
    # round 6: the headline gathers X on rank 0 (the N = 1 step's work), the distributed-X step is timed beside it; B is row-sharded;
    # the line says how many ranks the collective library spans (an all-reduce of ones)
    assert sh["ms_per_step_x_distributed"] > 0 and sh["value_x_distributed"] > 0 and "gathered" in sh["x"] and "row-sharded" in sh["b"]
    assert sh["rccl_ranks"] == 2 and sh["world_size"] == 2 and sh["collective_backend"] == "gloo"
    # the plan carries the flop bound AND the time bounds built from one unsharded step's per-level event times
    pl = sh["plan"]
    assert 1.0 <= pl["time_bound_speedup_latency"] <= pl["time_bound_speedup_share"] <= 2.0 + 1e-9 and pl["flop_bound_speedup"] <= 2.0 + 1e-9
    assert pl["measured_ms_one_gpu"] > 0 and len(pl["top_levels"]) == len(pl["top_levels_ms"]) >= 1
    # every panel is stored on exactly one rank
    mem = sh["per_rank_hbm_bytes"]
    assert mem["max_factor_panels"] < 0.75 * mem["sum_factor_panels"]
    # the additional cfg-4 line (3-D mesh, here 14^3 nodes), sharded the same way
    c4 = d["cfg4_sharded"]
    assert "error" not in c4 and c4["n"] == 14 ** 3 and c4["check"]["rel_residual"] < 1e-10 and c4["check"]["info"] == 0 and c4["value"] > 0


def test_sharded_run_that_hangs_exits_non_zero_with_the_replica_line():
    """rc == 0 only when the sharded run finished: a watchdog that fires prints the replica line ONCE with the error recorded
    and every rank (so the launcher too) leaves with a non-zero code."""
    d = _run(["--gpus", "2", "--rehearse", "--grid", "200", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-logpdf", "--no-cfg3",
              "--shard-timeout", "0.001"], expect_rc=1)      # torch.distributed.run maps any failed rank to exit code 1
    assert d["scaling"] == "weak" and "error" in d["sharded"] and "did not finish" in d["sharded"]["error"]
    assert d["replicas"]["value"] == d["value"]


def test_cfg4_line_is_skipped_and_says_so_when_the_deadline_leaves_no_room():
    """round 6: the line is printed behind the additional cfg-4 leg, so the leg only starts when --cfg4-budget seconds of --deadline are
    left (the driver ends a bench run after 600 s; 126^3 nodes cost ~170 s of host work per rank before the first kernel): the cfg-2
    headline stays, the skip is recorded, rc == 0."""
    d = _run(["--gpus", "2", "--rehearse", "--grid", "200", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-logpdf", "--no-cfg3",
              "--cfg4-grid", "14", "--cfg4-budget", "100000"])
    assert d["scaling"] == "strong" and "error" not in d["sharded"]
    assert d["cfg4_sharded"]["status"].startswith("skipped") and "--deadline" in d["cfg4_sharded"]["status"]
