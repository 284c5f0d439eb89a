#!/bin/bash
# per-launch durations of the contribution-block SYRK for two settings of an environment variable
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
var=$1; shift
for v in "$@"; do
  export $var=$v
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sl_$v -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 --no-logpdf --no-cfg3 > gpurun_out/sl_$v.json 2> gpurun_out/sl_$v.err || { tail -5 gpurun_out/sl_$v.err; exit 1; }
  python3 tools/timeline.py gpurun_out/sl_$v 0 | grep syrk > gpurun_out/sl_$v.txt
  echo "== $var=$v"; cat gpurun_out/sl_$v.txt
done
