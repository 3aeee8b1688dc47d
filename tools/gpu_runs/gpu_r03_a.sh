#!/bin/bash
# Round 3, GPU call A (after merging r03-k3-rotation-table): instruction-rate microbenchmark #2, the reference-vUv runs round 2 could not
# afford, the K3 rotation table's effect on the device, the whole -m gpu suite, per-kernel times of the merged build.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_runs/gpu_r03_a.sh'
set -x
O=gpurun_out/r03_a; mkdir -p $O
timeout 120 tools/microbench/bin/valu_rates2 > $O/valu_rates2.txt 2>&1; tail -5 $O/valu_rates2.txt
timeout 200 python tools/quick_time.py 3840 2160 20 > $O/quick_time_4k.txt 2>&1; cat $O/quick_time_4k.txt
export RFX_TEST_UV_REFERENCE=1 RFX_TEST_UNSEEN=1
timeout 400 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -s -v -k "vUv" --durations=5 > $O/uv_reference_1080p.txt 2>&1
grep -E "PASSED|FAILED|passed|failed|UNEXPLAINED [1-9]" $O/uv_reference_1080p.txt | tail -20
timeout 200 python -m pytest tests/test_node_host.py tests/test_zz_gpu_cube_environment.py -m gpu -q -k "cube" > $O/cube.txt 2>&1; tail -3 $O/cube.txt
unset RFX_TEST_UV_REFERENCE RFX_TEST_UNSEEN
timeout 200 python tools/gpu_runs/uv_model_check.py > $O/uv_model_check.txt 2>&1; tail -40 $O/uv_model_check.txt
timeout 400 python tools/parity_configs.py --impl hip --size 3840x2160 --steps 20 --it 1 --frames 2 --uv-model reference_gl --out $O/parity_configs2_reference_uv.txt > /dev/null 2>&1
tail -14 $O/parity_configs2_reference_uv.txt
timeout 500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -3
