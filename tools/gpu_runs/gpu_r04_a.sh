#!/bin/bash
# round 4, call A: K1 march rewrite (cs == 1 fold, live factor, wave-uniform loops) against the round-3 kernel and its knobs; -fno-slp-vectorize;
# the 4K streamed-vs-uploaded test; a bench line
mkdir -p gpurun_out/r04_a
cd "$GRAFT_REPO_ROOT"
( timeout 600 bash tools/time_variants.sh ) > gpurun_out/r04_a/variants.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streamed" > gpurun_out/r04_a/streamed.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r04_a/bench.txt 2>&1
tail -3 gpurun_out/r04_a/streamed.txt; tail -1 gpurun_out/r04_a/bench.txt | cut -c1-400
