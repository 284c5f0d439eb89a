#!/bin/bash
# Which unit is each kernel of a bench step bound by? Two counter passes over `bench.py --steps 1 --warmup 1`
# (each pass in its own run, counters only): address unit (TA) busy cycles and MFMA busy cycles next to the CU busy
# cycles, and the vector memory instruction counts. Output: gpurun_out/pmc_units.txt (per kernel: TA busy / CU busy,
# MFMA busy / (4 x CU busy), cycles of TA time per vector memory instruction).
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/pu1 gpurun_out/pu2
rocprofv3 --pmc TA_TA_BUSY_sum SQ_BUSY_CU_CYCLES -d gpurun_out/pu1 --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logpdf --no-cfg3 > /dev/null 2> gpurun_out/pu1.err || { tail -3 gpurun_out/pu1.err; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d gpurun_out/pu2 --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logpdf --no-cfg3 > /dev/null 2> gpurun_out/pu2.err || { tail -3 gpurun_out/pu2.err; exit 1; }
python3 - <<'PY' > gpurun_out/pmc_units.txt
import csv, glob, os
def load(d):
    f = sorted(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
    agg = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('gmrfx::', '').replace('void ', '')
        agg.setdefault(k, {}).setdefault(r['Counter_Name'], 0.0)
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    return agg
a, b = load('gpurun_out/pu1'), load('gpurun_out/pu2')
print(f"{'kernel':32s} {'CU busy cyc':>12s} {'TA busy':>8s} {'MFMA busy':>9s} {'VMEM rd':>10s} {'VMEM wr':>10s} {'TA cyc/VMEM':>11s} {'LDS inst':>10s}")
for k, d in sorted(a.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CU_CYCLES', 0)):
    cu = d.get('SQ_BUSY_CU_CYCLES', 0.0); ta = d.get('TA_TA_BUSY_sum', 0.0)
    e = b.get(k, {})
    vm = e.get('SQ_INSTS_VMEM_RD', 0.0) + e.get('SQ_INSTS_VMEM_WR', 0.0)
    if cu <= 0: continue
    print(f"{k[:32]:32s} {cu:12.3e} {ta/cu:8.2f} {e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)/(4*cu):9.2f} {e.get('SQ_INSTS_VMEM_RD', 0.0):10.3e} "
          f"{e.get('SQ_INSTS_VMEM_WR', 0.0):10.3e} {ta/max(vm,1):11.1f} {e.get('SQ_INSTS_LDS', 0.0):10.3e}")
PY
cat gpurun_out/pmc_units.txt
