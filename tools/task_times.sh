#!/bin/bash
# Average duration of the sweep-task kernels (sweep_chunk.hip: k_fwd_chunks / k_bwd_chunks / k_pack_diag; one wave per task and
# 16 columns: k_wave_task) in a 64-RHS solve at cfg 2, once per environment setting given ("" = defaults):
#     tools/task_times.sh "" "GMRFX_TASK_MODE=wave" ...
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
[ $# -eq 0 ] && set -- ""
for setting in "$@"; do
  rm -rf gpurun_out/td
  ( for kv in $setting; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/td -- python3 tools/sweep_levels.py run > /dev/null 2> gpurun_out/td.err ) || { tail -3 gpurun_out/td.err; exit 1; }
  f=$(ls gpurun_out/td/*/*kernel_stats.csv | head -1)
  echo "[${setting:-defaults}] $(python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    n = r['Name'].replace('void gmrfx::', '').replace('gmrfx::', '').split('(')[0]
    if any(k in n for k in ('k_fwd_chunks', 'k_bwd_chunks', 'k_pack_diag', 'k_wave_task')): print(n, 'calls', r['Calls'], 'avg_us=%.1f' % (float(r['AverageNs'])/1e3), 'min %.1f' % (float(r['MinNs'])/1e3), end='   ')
")"
done
