#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root through gpurun):
#   1. the default bench line                       -> gpurun_out/fp_line.json
#   2. rocprofv3 --kernel-trace --stats of bench.py -> gpurun_out/fp_trace/ (+ per-step summary)
#   3. two PMC passes (FETCH_SIZE, WRITE_SIZE), each in its own run, no tracing flags besides the counters
# Copy afterwards (on the build machine): see the cp lines at the end of this script's output.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
R=${1:-r01}
python3 bench.py --extras > gpurun_out/fp_line.json 2> gpurun_out/fp_line.err
echo "bench line done"
rm -rf gpurun_out/fp_trace gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fp_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io > gpurun_out/fp_trace.json 2> gpurun_out/fp_trace.err
python3 tools/prof_summary.py gpurun_out/fp_trace 12 > gpurun_out/fp_summary.txt      # 2 warm-up + 5 pipelined + 5 separate-call steps
cp $(ls -t gpurun_out/fp_trace/*/*kernel_stats.csv | head -1) gpurun_out/fp_kernel_stats.csv
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io > /dev/null 2> gpurun_out/pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io > /dev/null 2> gpurun_out/pmc_write.err
echo "pmc write done"
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/fp_pmc_traffic.json 1000 64
cp $(ls -t gpurun_out/pmc_fetch/*/*counter_collection.csv | head -1) gpurun_out/fp_pmc_fetch_size.csv
cp $(ls -t gpurun_out/pmc_write/*/*counter_collection.csv | head -1) gpurun_out/fp_pmc_write_size.csv
rm -rf gpurun_out/fp_trace/*/*kernel_trace.csv.bak
head -12 gpurun_out/fp_summary.txt
tail -c 700 gpurun_out/fp_line.json
