#!/bin/bash
# Where the selected inversion's dense kernel waits (VERDICT r05 "Next round" #6): SQ wait / issue counters and unit busy fractions of
# every kernel of the selected inversion of cfg 3, per tree level (the library's level marks cut the counter passes), product library.
# Output: gpurun_out/<tag>_cfg3_selinv_stalls.txt (tag: $1, default r06); copy to profiles/.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
rm -rf gpurun_out/c3_sq1 gpurun_out/c3_sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/c3_sq1 -- python3 tools/cfg3_profile.py run > gpurun_out/c3_sq1.log 2>&1 || { tail -3 gpurun_out/c3_sq1.log; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES TA_TA_BUSY_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d gpurun_out/c3_sq2 -- python3 tools/cfg3_profile.py run > gpurun_out/c3_sq2.log 2>&1 || { tail -3 gpurun_out/c3_sq2.log; exit 1; }
python3 - "$TAG" <<'PY'
import csv, glob, os, sys, collections
tag = sys.argv[1]
def load(d):
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    # dispatches in order, each with its counters
    disp = collections.OrderedDict()
    for r in rows:
        d = disp.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name'].split('(')[0].replace('void ', '').replace('gmrfx::', ''),
                                               'wg': int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 0)) or 0), 'grid': int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0), 'c': {}})
        d['c'][r['Counter_Name']] = d['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return list(disp.values())
def by_level(disp):
    """the LAST selected inversion of the run: phase-4 marks (256 threads), grid / 256 - 2 = level"""
    marks = [i for i, d in enumerate(disp) if d['name'].startswith('k_level_mark') and d['wg'] == 256]
    # two inversions per run: the marks of the last one = the last (number of levels) marks
    lv = [disp[i]['grid'] // 256 - 2 for i in marks]
    nl = max(lv) + 1
    marks = marks[-nl:]
    out = collections.OrderedDict()
    for k, i in enumerate(marks):
        end = marks[k + 1] if k + 1 < len(marks) else len(disp)
        level = disp[i]['grid'] // 256 - 2
        agg = out.setdefault(level, {})
        for d in disp[i + 1:end]:
            if d['name'].startswith('k_level_mark') or not d['name'].startswith('k_sel'):
                if not d['name'].startswith(('k_sel', 'k_trsm')): 
                    if d['name'].startswith('k_level_mark'): break
                    continue
            a = agg.setdefault(d['name'], collections.defaultdict(float))
            for c, v in d['c'].items(): a[c] += v
            a['launches'] += 1
    return out
A, B = by_level(load('gpurun_out/c3_sq1')), by_level(load('gpurun_out/c3_sq2'))
lines = ["# tools/cfg3_sq.sh: selected inversion of cfg 3 (10^6 nodes), per tree level (top-down) and kernel -- SQ counters of the product library",
         "# parked = SQ_WAIT_ANY (wave at s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY (issue stall: MFMA operand / pipe), active = SQ_ACTIVE_INST_ANY, LDS = SQ_WAIT_INST_LDS; shares of SQ_WAVE_CYCLES",
         "# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), TA busy = TA_TA_BUSY_sum / SQ_BUSY_CU_CYCLES (second pass)",
         f"{'level':>5s} {'kernel':28s} {'launches':>8s} {'wave cycles':>12s} {'parked':>7s} {'stall':>6s} {'active':>7s} {'LDS':>5s} | {'MFMA busy':>9s} {'TA busy':>8s} {'VMEM rd':>10s} {'LDS inst':>10s}"]
for lv in A:
    for k, a in sorted(A[lv].items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
        wc = a.get('SQ_WAVE_CYCLES', 0.0)
        if wc <= 0: continue
        b = B.get(lv, {}).get(k, {})
        cu = b.get('SQ_BUSY_CU_CYCLES', 0.0)
        lines.append(f"{lv:5d} {k[:28]:28s} {int(a['launches']):8d} {wc:12.3e} {a.get('SQ_WAIT_ANY', 0)/wc:7.2f} {a.get('SQ_WAIT_INST_ANY', 0)/wc:6.2f} {a.get('SQ_ACTIVE_INST_ANY', 0)/wc:7.2f} {a.get('SQ_WAIT_INST_LDS', 0)/wc:5.2f} | "
                     f"{(b.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)/(4*cu) if cu else 0):9.2f} {(b.get('TA_TA_BUSY_sum', 0)/cu if cu else 0):8.2f} {b.get('SQ_INSTS_VMEM_RD', 0):10.3e} {b.get('SQ_INSTS_LDS', 0):10.3e}")
open(f'gpurun_out/{tag}_cfg3_selinv_stalls.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
