#!/usr/bin/env python3
"""A/B of environment switches on the cfg-2 bench step: `python3 tools/ab_env.py "A=1 B=2" "A=0" ...` runs bench.py (short, no extras)
once per setting in a child process and prints ms_per_step (pipelined), separate-calls step and factor time. "" = defaults."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for setting in sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-logpdf",
                        "--no-cfg3", "--no-host-io"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        pp = d.get("pipelined_phases_ms", {})
        print(f"{setting or '(defaults)':40s} step {d['ms_per_step']:.3f} ms | separate calls {d['ms_per_step_separate_calls']:.3f} (factor {d['phases_ms']['factor']:.3f}) | "
              f"pipelined factor {pp.get('factor', float('nan')):.3f} + {pp.get('behind_factor', float('nan')):.3f} | equal {d['check'].get('pipelined_equals_separate')}", flush=True)
    except Exception as e:
        print(f"{setting}: failed ({e!r}); stderr tail: {r.stderr[-300:]}", flush=True)
