#!/bin/bash
# A/B a tuning/experiment environment variable on the default bench: tools/ab_env.sh VAR v1 v2 ...
var=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { tail -5 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("$var=$v","step",round(d["ms_per_step"],3),"factor",round(d["phases_ms"]["factor"],3),"solve",round(d["phases_ms"]["solve"],3),"fwd",round(d["phases_ms"]["solve_fwd"],3),"bwd",round(d["phases_ms"]["solve_bwd"],3),"syrk",round(d["roofline"]["ms_per_step"],3),"TF",round(d["roofline"]["achieved"],2),"resid",d["check"]["rel_residual"],"logdet",d["check"]["logdet"])
PY
done
