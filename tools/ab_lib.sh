#!/bin/bash
# A/B of two builds of the library on the default bench: gpurun_out/ab_lib.sh tag1=path1 tag2=path2 ...
for kv in "$@"; do
  tag=${kv%%=*}; lib=${kv#*=}
  if [ "$lib" = "default" ]; then unset GMRFX_LIB; else export GMRFX_LIB=$lib; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-io --steps 20 --warmup 3 > gpurun_out/abl_$tag.json 2> gpurun_out/abl_$tag.err || { tail -5 gpurun_out/abl_$tag.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$tag.json"))
print("$tag","step",round(d["ms_per_step"],3),"sep",round(d["ms_per_step_separate_calls"],3),"factor",round(d["phases_ms"]["factor"],3),"syrk",round(d["roofline"]["ms_per_step"],3),"selinv",round(d["phases_ms"]["ms_selinv"],3),"rand256",round(d["phases_ms"]["ms_rand256"],3),"resid",d["check"]["rel_residual"],"logdet",d["check"]["logdet"])
PY
done
