#!/bin/bash
# First GPU call of round 3 (run from the repo root on the GPU box):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_runs/gpu_r03_first.sh'
# 1. the BASELINE-config runs under rfx_set_uv_model(RFX_UV_REFERENCE_GL) that round 2 could not afford (1080p configs[1] and the
#    configs[4] options; expect nothing UNEXPLAINED and an order of magnitude fewer K3 flips than under the ideal vUv);
# 2. the 4K whole-frame report under the reference vUv (profiles/r03_parity/parity_configs2_reference_uv.txt);
# 3. the whole -m gpu suite and the bench line, so that the round starts from a known-green state.
# NOTE `-k`: the test FUNCTION is called ..._vs_reference_glsl, so `-k reference_gl` selects every parametrisation (round 2 lost its
# last GPU minute to that); the ids of the reference-vUv cases contain "vUv".
set -x
mkdir -p gpurun_out/r03_a
export RFX_TEST_UV_REFERENCE=1 RFX_TEST_UNSEEN=1   # UNSEEN: tests written after round 2's GPU budget (Node host with a cube environment)
timeout 300 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -s -v -k "vUv" --durations=5 > gpurun_out/r03_a/uv_reference_1080p.txt 2>&1
timeout 200 python -m pytest tests/test_node_host.py tests/test_zz_gpu_cube_environment.py -m gpu -q -k "cube" > gpurun_out/r03_a/cube.txt 2>&1; tail -3 gpurun_out/r03_a/cube.txt
grep -E "PASSED|FAILED|passed|failed|UNEXPLAINED [1-9]" gpurun_out/r03_a/uv_reference_1080p.txt | tail -20
timeout 400 python tools/parity_configs.py --impl hip --size 3840x2160 --steps 20 --it 1 --frames 2 --uv-model reference_gl --out gpurun_out/r03_a/parity_configs2_reference_uv.txt > /dev/null 2>&1
tail -12 gpurun_out/r03_a/parity_configs2_reference_uv.txt
unset RFX_TEST_UV_REFERENCE RFX_TEST_UNSEEN
timeout 600 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r03_a/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/r03_a/pytest_gpu.log | tail -3
timeout 300 python bench.py > gpurun_out/r03_a/bench.json 2> gpurun_out/r03_a/bench.err
cat gpurun_out/r03_a/bench.json
