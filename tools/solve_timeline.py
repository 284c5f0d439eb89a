#!/usr/bin/env python3
"""Kernel timeline of the LAST solve (between its two k_permute launches) from a rocprofv3 kernel trace.
usage: tools/solve_timeline.py <rocprof output dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nm = lambda r: r['Kernel_Name'].split('(')[0].replace('gmrfx::', '').replace('void ', '')
idx = [i for i, r in enumerate(rows) if nm(r).startswith('k_permute')]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
agg = {}
for r in rows[a:b + 1]:
    st, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    n = nm(r)
    print(f"{(st-t0)/1e3:8.1f} {(e-st)/1e3:7.1f} wg {g:6d} {n} q{r['Queue_Id']}")
    agg[n] = agg.get(n, 0) + (e - st) / 1e3
print({k: round(v) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])})
