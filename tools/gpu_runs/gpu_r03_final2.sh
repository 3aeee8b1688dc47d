#!/bin/bash
# Round 3, final call 2: the profile collection at the benchmarked commit (bench line with the llvmpipe baseline, kernel trace of bench.py,
# nine counter passes over the torch-free target), then the same-box bench line again.
set -x
PMC_TIMEOUT=120 bash tools/collect_profiles.sh r03_final
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r03_final/bench_again.json 2> /dev/null; cut -c1-400 gpurun_out/r03_final/bench_again.json
