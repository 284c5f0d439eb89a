#!/bin/bash
# Per-level profile of the triangular sweeps (tools/sweep_levels.py): one kernel trace + two counter passes, then the table.
# usage (GPU box, repo root): tools/sweep_profile.sh [tag]   -> gpurun_out/<tag>_sweep_levels.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
T=${1:-swl}
mkdir -p gpurun_out
rm -rf gpurun_out/${T}_t gpurun_out/${T}_f gpurun_out/${T}_w
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_t -- python3 tools/sweep_levels.py run > gpurun_out/${T}_run.json 2> gpurun_out/${T}_t.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${T}_f --output-format csv -- python3 tools/sweep_levels.py run > /dev/null 2> gpurun_out/${T}_f.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${T}_w --output-format csv -- python3 tools/sweep_levels.py run > /dev/null 2> gpurun_out/${T}_w.err
echo "write done"
python3 tools/sweep_levels.py join gpurun_out/${T}_t gpurun_out/${T}_f gpurun_out/${T}_w > gpurun_out/${T}_sweep_levels.txt
python3 tools/solve_timeline.py gpurun_out/${T}_t > gpurun_out/${T}_solve_timeline.txt
rm -rf gpurun_out/${T}_f gpurun_out/${T}_w
cat gpurun_out/${T}_sweep_levels.txt
