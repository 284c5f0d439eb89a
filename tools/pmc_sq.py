#!/usr/bin/env python3
"""Aggregates one `rocprofv3 --pmc <SQ counters>` run per kernel name: sums of every counter over the dispatches.
usage: pmc_sq.py <rocprof output dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*counter_collection.csv'), key=os.path.getmtime)[-1]
agg, cnt = {}, {}
names = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('gmrfx::', '').replace('void ', '')
    c = r['Counter_Name']; v = float(r['Counter_Value'])
    names.add(c)
    agg.setdefault(k, {}).setdefault(c, 0.0)
    agg[k][c] += v
    cnt[(k, c)] = cnt.get((k, c), 0) + 1
names = sorted(names)
print("kernel".ljust(30), "n".rjust(6), " ".join(n[-22:].rjust(22) for n in names))
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get(names[0], 0))):
    print(k[:30].ljust(30), str(cnt[(k, names[0])]).rjust(6), " ".join(f"{d.get(n, 0):22.3e}" for n in names))
