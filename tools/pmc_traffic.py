#!/usr/bin/env python3
"""Aggregate the HBM-traffic PMC passes into profiles/<name>.json.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json 1000 64

Counter unit: KB. FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of the wide
coalesced reads). Only the dispatches of the LAST bench step are counted (from its first
factorisation kernel on; the untimed extras after the step -- logdet -- are tiny)."""
import csv, glob, json, sys, collections

def load(d):
    import os
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f[-1])))    # newest run in the directory
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    names = [r["Kernel_Name"].split("(")[0].replace("gmrfx::", "").replace("void ", "") for r in rows]
    start = last_step_start(names)
    return [(n, float(r["Counter_Value"])) for n, r in zip(names[start:], rows[start:])]

def last_step_start(names):
    """index of the first factorisation kernel of the LAST refactorise+solve step"""
    prev_factor, start = False, 0
    for i, n in enumerate(names):
        if not n.startswith("k_"): continue          # runtime copy / fill kernels
        f = phase(n) == "factor"
        if f and not prev_factor: start = i
        prev_factor = f
    return start

def phase(n):
    if n.startswith(("k_factor", "k_assemble", "k_potrf", "k_trsm", "k_gemm_nt", "k_syrk")): return "factor"
    if n.startswith(("k_inv_stage", "k_pack_diag", "k_rdiag")): return "dense_inverse"      # once per factorisation, on the first solve
    if n.startswith(("k_fwd", )): return "sweep_forward"
    if n.startswith(("k_bwd", )): return "sweep_backward"
    if n.startswith("k_permute"): return "permute"
    return None

def main():
    fetch, write, out, grid, nrhs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    F, W = load(fetch), load(write)
    per = collections.defaultdict(lambda: {"fetch_x2": 0.0, "write": 0.0, "launches": 0})
    res = {k: {"fetch_bytes": 0.0, "write_bytes": 0.0} for k in ("factor", "dense_inverse", "sweep_forward", "sweep_backward", "permute")}
    # k_xmul / k_copy_own belong to the sweep that is running: forward until the first k_bwd kernel shows up
    def walk(L, key, scale):
        seen_bwd = False
        for n, v in L:
            if n.startswith("k_bwd"): seen_bwd = True
            ph = phase(n)
            if ph is None and n.startswith(("k_xmul", "k_copy_own")): ph = "sweep_backward" if seen_bwd else "sweep_forward"
            b = v * 1024.0 * scale
            per[n][key] += b / 1e9
            if key == "fetch_x2": per[n]["launches"] += 1
            if ph: res[ph]["fetch_bytes" if key == "fetch_x2" else "write_bytes"] += b
    walk(F, "fetch_x2", 2.0)
    walk(W, "write", 1.0)
    for v in res.values(): v["total_bytes"] = v["fetch_bytes"] + v["write_bytes"]
    res["per_kernel_GB_per_step"] = {k: {kk: (round(vv, 3) if kk != "launches" else vv) for kk, vv in v.items()} for k, v in sorted(per.items())}
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io); counter unit KB; "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); values per bench step "
                   "(refactorise + solve).")
    res["workload"] = {"grid": grid, "nrhs": nrhs}
    # the source tree these counters belong to: bench.py quotes the figures only while it is timing the same tree
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianmarkovrandomfields.jl_amd"))
    from gmrfx._lib import source_tree_hash
    res["csrc_hash"] = source_tree_hash()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("per_kernel_GB_per_step", "note")}))

if __name__ == "__main__":
    main()
