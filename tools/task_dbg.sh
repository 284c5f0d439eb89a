#!/bin/bash
# Timing experiment on the sweep-task kernels: average duration of k_fwd_task / k_bwd_task for combinations of
# GMRFX_TASK_NC (64 / 32 columns per workgroup) and GMRFX_TASK_DBG (1 = no front loop, 2 = no L2 warm-up, 3 = neither).
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
for nc in ${NCS:-64 32}; do for dbg in ${DBGS:-0 1 2 3}; do
  export GMRFX_TASK_NC=$nc GMRFX_TASK_DBG=$dbg
  rm -rf gpurun_out/td
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/td -- python3 tools/sweep_levels.py run > /dev/null 2> gpurun_out/td.err || { tail -3 gpurun_out/td.err; exit 1; }
  f=$(ls gpurun_out/td/*/*kernel_stats.csv | head -1)
  echo "NC=$nc DBG=$dbg $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_fwd_task' in r['Name'] or 'k_bwd_task' in r['Name'] or 'k_wave_task' in r['Name']: print(r['Name'][12:32], 'calls', r['Calls'], 'avg_us=%.1f' % (float(r['AverageNs'])/1e3), 'min %.1f max %.1f' % (float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3), end='   ')
")"
done; done
