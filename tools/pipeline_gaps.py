#!/usr/bin/env python3
"""What does the factorisation pay for the forward sweep beside it in the pipelined call -- waiting for a dispatch slot (GAPS between
dependent launches grow) or loaded memory latency (DURATIONS grow)? (VERDICT r05 "Next round" #5.)

From two kernel traces of bench.py steps (program directly behind `--`):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pg_sep -- python3 bench.py --steps 4 --warmup 2 --separate-calls --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pg_pip -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io
    python3 tools/pipeline_gaps.py gpurun_out/pg_sep gpurun_out/pg_pip > gpurun_out/r06_pipeline_gaps.txt
For the panel-chain kernels of the top of the tree (the latency variants: k_potrf64_b<8, 4>, k_trsm<0, 1>, k_gemm_nt<1>) of every timed
step: launches, mean / summed DURATION, and mean / summed GAP = start - end of the previous kernel of the same stream. Under the
profiler the host is slower than the GPU at the top of the tree, so gaps are inflated in BOTH traces alike; what counts is the difference.
"""
import csv, glob, os, sys
import numpy as np

CHAIN = ("k_potrf64_b<8, 4>", "k_trsm<0, 1>", "k_gemm_nt<1>", "k_gemm_nt<2>", "k_trsm<0, 0>", "k_syrk_cb_rec", "k_assemble_lds<1>", "k_assemble_lds<0>")


def nm(r):
    return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gmrfx::", "")


def load(d):
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    return rows


def steps(rows):
    """one step = from a k_gather_values (first kernel of a factorisation) to the next; the last 3 complete ones"""
    idx = [i for i, r in enumerate(rows) if nm(r) == "k_gather_values"]
    segs = [(idx[k], idx[k + 1]) for k in range(len(idx) - 1)]
    return segs[-3:]


def analyse(rows, seg):
    a, b = seg
    part = rows[a:b]
    # the main stream of the factorisation = the stream of its k_potrf64_b launches with the most launches
    cnt = {}
    for r in part:
        if nm(r).startswith("k_potrf64_b"):
            cnt[r["Stream_Id"]] = cnt.get(r["Stream_Id"], 0) + 1
    streams = sorted(cnt, key=lambda k: -cnt[k])[:2]          # two panel chains
    out = {}
    last_end = {}
    fact_end = max(r["e"] for r in part if nm(r).startswith(("k_potrf64_b", "k_syrk_cb", "k_trsm", "k_gemm_nt")))
    t0 = part[0]["s"]
    for r in part:
        st = r["Stream_Id"]
        if st in streams and r["s"] <= fact_end:
            gap = r["s"] - last_end[st] if st in last_end else 0
            k = nm(r)
            if k in CHAIN:
                # top of the tree only for the bulk kernels: small grids
                wg = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * max(int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1), 1) * max(int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1), 1)
                late = (r["s"] - t0) > 0.55 * (fact_end - t0)          # the last ~45 % of the factorisation's span = levels ~12-17
                key = k + (" [top]" if late else " [below]")
                o = out.setdefault(key, {"n": 0, "dur": 0.0, "gap": 0.0, "wg": 0})
                o["n"] += 1; o["dur"] += (r["e"] - r["s"]) / 1e3; o["gap"] += max(gap, 0) / 1e3; o["wg"] += wg
        if st in streams:
            last_end[st] = max(last_end.get(st, 0), r["e"])
    side = [r for r in part if r["Stream_Id"] not in streams and r["s"] < fact_end and nm(r).startswith(("k_fwd", "k_xmul", "k_inv_stage", "k_pack_diag"))]
    return out, (fact_end - t0) / 1e3, len(side), sum(r["e"] - r["s"] for r in side) / 1e3


def main():
    dsep, dpip = sys.argv[1], sys.argv[2]
    res = {}
    for tag, d in (("separate calls", dsep), ("pipelined call", dpip)):
        rows = load(d)
        acc, spans, nside, tside = {}, [], [], []
        segs = steps(rows)
        for seg in segs:
            o, span, ns, ts = analyse(rows, seg)
            spans.append(span); nside.append(ns); tside.append(ts)
            for k, v in o.items():
                a = acc.setdefault(k, {"n": 0, "dur": 0.0, "gap": 0.0, "wg": 0})
                for q in v:
                    a[q] += v[q]
        for a in acc.values():
            for q in a:
                a[q] /= len(segs)
        res[tag] = (acc, float(np.mean(spans)), float(np.mean(nside)), float(np.mean(tside)))
    print("# tools/pipeline_gaps.py: panel-chain kernels of the factorisation, per step (mean of the last 3 traced steps), cfg 2; times in us")
    print("# [top] = launched in the last 45 % of the factorisation's span (levels ~12-17: the latency-bound chains), [below] = before")
    for tag in res:
        print(f"# {tag}: factorisation span under the profiler {res[tag][1]:.0f} us; forward-sweep / inverse kernels beside it: {res[tag][2]:.0f} launches, {res[tag][3]:.0f} us of kernel time")
    print(f"{'kernel':34s} | {'separate: n':>11s} {'dur/launch':>10s} {'gap/launch':>10s} {'sum dur':>8s} {'sum gap':>8s} | {'pipelined: n':>12s} {'dur/launch':>10s} {'gap/launch':>10s} {'sum dur':>8s} {'sum gap':>8s} | {'d dur':>7s} {'d gap':>7s}")
    keys = sorted(set(res["separate calls"][0]) | set(res["pipelined call"][0]), key=lambda k: (k.endswith("[below]"), k))
    tot = [0.0, 0.0]
    for k in keys:
        a = res["separate calls"][0].get(k, {"n": 0, "dur": 0, "gap": 0})
        b = res["pipelined call"][0].get(k, {"n": 0, "dur": 0, "gap": 0})
        dd, dg = b["dur"] - a["dur"], b["gap"] - a["gap"]
        if k.endswith("[top]"):
            tot[0] += dd; tot[1] += dg
        print(f"{k:34s} | {a['n']:11.0f} {a['dur'] / max(a['n'], 1):10.2f} {a['gap'] / max(a['n'], 1):10.2f} {a['dur']:8.0f} {a['gap']:8.0f} | "
              f"{b['n']:12.0f} {b['dur'] / max(b['n'], 1):10.2f} {b['gap'] / max(b['n'], 1):10.2f} {b['dur']:8.0f} {b['gap']:8.0f} | {dd:7.0f} {dg:7.0f}")
    print(f"# [top] kernels together: durations {tot[0]:+.0f} us, gaps {tot[1]:+.0f} us per step in the pipelined call")


if __name__ == "__main__":
    main()
