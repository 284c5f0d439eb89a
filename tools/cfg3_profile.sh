#!/bin/bash
# cfg-3 evidence on a GPU box (run from the repo root through gpurun): kernel trace + two PMC passes of tools/cfg3_profile.py,
# joined into gpurun_out/<tag>_cfg3_{kernel_stats.csv,pmc_traffic.json,selinv_levels.txt} (tag: first argument, default r05); copy them
# to profiles/ afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
R=${1:-r05}
rm -rf gpurun_out/c3_trace gpurun_out/c3_fetch gpurun_out/c3_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3_trace -- python3 tools/cfg3_profile.py run > gpurun_out/c3_trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/c3_fetch -- python3 tools/cfg3_profile.py run > gpurun_out/c3_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/c3_write -- python3 tools/cfg3_profile.py run > gpurun_out/c3_write.log 2>&1
echo "write done"
python3 tools/cfg3_profile.py join gpurun_out/c3_trace gpurun_out/c3_fetch gpurun_out/c3_write gpurun_out/${R}_cfg3
