#!/bin/bash
# kernel traces of the separate-calls step and of the pipelined step (program directly behind `--`), then tools/pipeline_gaps.py
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/pg_sep gpurun_out/pg_pip
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pg_sep -- python3 bench.py --steps 4 --warmup 2 --separate-calls --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io > gpurun_out/pg_sep.json 2> gpurun_out/pg_sep.err || { tail -3 gpurun_out/pg_sep.err; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pg_pip -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-logpdf --no-cfg3 --no-host-io > gpurun_out/pg_pip.json 2> gpurun_out/pg_pip.err || { tail -3 gpurun_out/pg_pip.err; exit 1; }
python3 tools/pipeline_gaps.py gpurun_out/pg_sep gpurun_out/pg_pip > gpurun_out/r06_pipeline_gaps.txt
cat gpurun_out/r06_pipeline_gaps.txt
