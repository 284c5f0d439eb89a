#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel time per bench step and the cost of the
small-grid (latency-bound) launches. usage: prof_summary.py <dir> [steps_incl_warmup]"""
import collections, csv, glob, sys
import numpy as np
d = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
import os
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]   # newest run in the directory
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('gmrfx::', '')
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    by[name].append((g, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
tot = sum(sum(t for _, t in v) for v in by.values())
print(f"{'kernel':28s} {'calls/step':>10s} {'ms/step':>9s} {'avg_us':>8s} {'small-grid(<=64 WG) calls/step, avg_us, ms/step'}")
for name, v in sorted(by.items(), key=lambda kv: -sum(t for _, t in kv[1])):
    t = np.array([x[1] for x in v]); g = np.array([x[0] for x in v])
    sm = g <= 64
    print(f"{name:28s} {len(v)/nsteps:10.1f} {t.sum()/nsteps/1e3:9.3f} {t.mean():8.1f}   {sm.sum()/nsteps:6.1f} {t[sm].mean() if sm.any() else 0:8.1f} {t[sm].sum()/nsteps/1e3:8.3f}")
print(f"total kernel time per step: {tot/nsteps/1e3:.2f} ms")
