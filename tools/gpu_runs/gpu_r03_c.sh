#!/bin/bash
# Round 3, GPU call C: what the instruction diet did to the bits (pre-diet kernels vs current, stage by stage at 4K), the whole default
# -m gpu suite with the reference vUv as the default and the 8K band case, the bench line with the measured stream copy.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_runs/gpu_r03_c.sh'
set -x
O=gpurun_out/r03_c; mkdir -p $O
timeout 300 python tools/diff_libs.py realism-effects_amd/csrc/variants/librfx_prediet_kernels.so realism-effects_amd/csrc/librfx_hip.so 3840x2160 2 > $O/diff_prediet_vs_diet_4k.txt 2>&1; cat $O/diff_prediet_vs_diet_4k.txt
timeout 900 python -m pytest tests -m gpu -q -x --durations=12 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^[0-9.]+s " $O/pytest_gpu.log | tail -16
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-1800 $O/bench.json
