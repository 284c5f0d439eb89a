#!/usr/bin/env python3
"""Print the kernel timeline of the LAST bench step from a rocprofv3 kernel trace (csv).
usage: tools/timeline.py <rocprof output dir> [min_us]"""
import csv, glob, sys
d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
import os
f = sorted(glob.glob(d + '/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('gmrfx::', '').replace('void ', '') for r in rows]
FACT = ('k_factor', 'k_assemble', 'k_potrf', 'k_trsm', 'k_gemm_nt', 'k_syrk')
prev, s = False, 0
for i, n in enumerate(names):
    if not n.startswith('k_'): continue
    f = n.startswith(FACT)
    if f and not prev: s = i
    prev = f
t0 = int(rows[s]['Start_Timestamp']); prev = t0
for r, n in zip(rows[s:], names[s:]):
    st, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    if (e - st) / 1e3 >= min_us:
        print(f"{(st-t0)/1e3:9.1f} {(e-st)/1e3:8.1f} gap {(st-prev)/1e3:6.1f} wg {g:6d} {n} q{r['Queue_Id']}")
    prev = e
