#!/bin/bash
# Evidence for the sweep-task kernels (VERDICT r05 "Next round" #1): (1) cycle stamps per chunk (tools/chunk_cycles.py on the
# -DGMRFX_CYC build), (2) SQ wait / issue counters and (3) L1->L2 read requests against L2-miss bytes for the kernels of one solve
# (counter passes in their own runs, product library). Output: gpurun_out/<tag>_chunk_cycles.txt (tag: $1, default r06).
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
python3 tools/chunk_cycles.py > gpurun_out/${TAG}_chunk_cycles.txt 2> gpurun_out/chunk_cycles.err || { tail -5 gpurun_out/chunk_cycles.err; exit 1; }
rm -rf gpurun_out/cc1 gpurun_out/cc2 gpurun_out/cc3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES -d gpurun_out/cc1 --output-format csv -- python3 tools/sweep_levels.py run > /dev/null 2> gpurun_out/cc1.err || { tail -3 gpurun_out/cc1.err; exit 1; }
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/cc2 --output-format csv -- python3 tools/sweep_levels.py run > /dev/null 2> gpurun_out/cc2.err || { tail -3 gpurun_out/cc2.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/cc3 --output-format csv -- python3 tools/sweep_levels.py run > /dev/null 2> gpurun_out/cc3.err || { tail -3 gpurun_out/cc3.err; exit 1; }
python3 - <<'PY' >> gpurun_out/${TAG}_chunk_cycles.txt
import csv, glob, os
def load(d):
    fs = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    agg, cnt = {}, {}
    if not fs: return agg, cnt
    seen = set()
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name'].split('(')[0].replace('gmrfx::', '').replace('void ', '')
        agg.setdefault(k, {}).setdefault(r['Counter_Name'], 0.0)
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (r['Dispatch_Id'], k) not in seen:
            seen.add((r['Dispatch_Id'], k)); cnt[k] = cnt.get(k, 0) + 1
    return agg, cnt
a, na = load('gpurun_out/cc1'); b, nb = load('gpurun_out/cc2'); c, nc = load('gpurun_out/cc3')
print("\n== SQ counters per kernel of the solve (tools/sweep_levels.py run: 3 solves at cfg 2, 64 right-hand sides; sums over its launches), product library")
print("   WAIT_ANY = wave parked (s_waitcnt / barrier), WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; shares of SQ_WAVE_CYCLES")
print(f"{'kernel':30s} {'launches':>8s} {'wave cycles':>12s} {'parked':>7s} {'issue stall':>11s} {'active':>7s} {'LDS stall':>9s}")
for k, d in sorted(a.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    wc = d.get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0 or not k.startswith('k_'): continue
    print(f"{k[:30]:30s} {na.get(k, 0):8d} {wc:12.3e} {d.get('SQ_WAIT_ANY', 0)/wc:7.2f} {d.get('SQ_WAIT_INST_ANY', 0)/wc:11.2f} {d.get('SQ_ACTIVE_INST_ANY', 0)/wc:7.2f} {d.get('SQ_WAIT_INST_LDS', 0)/wc:9.2f}")
print("\n== L1 -> L2 read requests against L2-miss bytes (how many times a byte enters a compute unit): TCP_TCC_READ_REQ x 64 B (128 B lines count once per 64 B half),")
print("   FETCH_SIZE x 2 (gfx950 correction, KB -> bytes); L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS)")
print(f"{'kernel':30s} {'launches':>8s} {'L1->L2 GB':>10s} {'L2-miss GB':>10s} {'ratio':>6s} {'L2 hit':>7s} {'L1 accesses':>12s}")
for k, d in sorted(b.items(), key=lambda kv: -kv[1].get('TCP_TCC_READ_REQ_sum', 0)):
    if not k.startswith('k_'): continue
    rq = d.get('TCP_TCC_READ_REQ_sum', 0.0) * 64.0
    fs = c.get(k, {}).get('FETCH_SIZE', 0.0) * 1024.0 * 2.0
    h, m = d.get('TCC_HIT_sum', 0.0), d.get('TCC_MISS_sum', 0.0)
    print(f"{k[:30]:30s} {nb.get(k, 0):8d} {rq/1e9:10.3f} {fs/1e9:10.3f} {rq/max(fs,1):6.2f} {h/max(h+m,1):7.2f} {d.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):12.3e}")
PY
cat gpurun_out/${TAG}_chunk_cycles.txt
