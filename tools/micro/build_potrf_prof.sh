#!/bin/bash
# builds potrf_prof (timing + checks) and potrf_prof_cyc (with -DGMRFX_CYC: cycle stamps) next to this script
set -e
cd "$(dirname "$0")"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -DGMRFX_CYC -I../../gaussianmarkovrandomfields.jl_amd/csrc potrf_prof.hip -o potrf_prof_cyc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -I../../gaussianmarkovrandomfields.jl_amd/csrc potrf_prof.hip -o potrf_prof
