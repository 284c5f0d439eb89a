#!/bin/bash
# Average duration of the sweep-task kernels (k_fwd_task / k_bwd_task: 16-wave workgroup form; k_wave_task: one wave per
# task and 16 columns) in a 64-RHS solve at cfg 2, for GMRFX_TASK_MODE in "$@" (default: wg wave).
# (The timing switches that located the bottleneck in round 3 -- front loop / warm-up / operand loads compiled out -- are
#  gone from the kernels: they cost 0.1 ms per sweep in register pressure. Results: DESIGN.md section 5.)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
for mode in ${@:-wg wave}; do
  export GMRFX_TASK_MODE=$mode
  rm -rf gpurun_out/td
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/td -- python3 tools/sweep_levels.py run > /dev/null 2> gpurun_out/td.err || { tail -3 gpurun_out/td.err; exit 1; }
  f=$(ls gpurun_out/td/*/*kernel_stats.csv | head -1)
  echo "MODE=$mode $(python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'k_fwd_task' in r['Name'] or 'k_bwd_task' in r['Name'] or 'k_wave_task' in r['Name']: print(r['Name'][12:32], 'calls', r['Calls'], 'avg_us=%.1f' % (float(r['AverageNs'])/1e3), 'min %.1f max %.1f' % (float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3), end='   ')
")"
done
