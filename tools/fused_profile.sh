#!/bin/bash
# Per-kernel times of the separate and the pipelined factor + solve step (rocprofv3 kernel trace): which kernels of the
# factorisation get slower when the forward sweep runs beside them. Usage (GPU box): bash tools/fused_profile.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in separate pipelined; do
  rm -rf /tmp/fp_$mode
  GMRFX_PROFILE_MODE=$mode rocprofv3 --kernel-trace --stats -d /tmp/fp_$mode -o out --output-format csv -- python3 $R/tools/fused_step_one.py $mode > /tmp/fp_$mode.log 2>&1 || { tail -5 /tmp/fp_$mode.log; exit 1; }
  f=$(find /tmp/fp_$mode -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/fused_${mode}_kernel_stats.csv
done
python3 - <<PY
import csv
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r["Name"]]=(int(r["Calls"]), float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3)
    return d
a=load("$R/gpurun_out/fused_separate_kernel_stats.csv"); b=load("$R/gpurun_out/fused_pipelined_kernel_stats.csv")
print(f"{'kernel':44s} {'calls':>7s} {'sep ms':>9s} {'pipe ms':>9s} {'sep us':>8s} {'pipe us':>8s}")
for k in sorted(a, key=lambda k: -a[k][1])[:28]:
    if k in b: print(f"{k[:44]:44s} {a[k][0]:7d} {a[k][1]:9.2f} {b[k][1]:9.2f} {a[k][2]:8.1f} {b[k][2]:8.1f}")
PY
