#!/bin/bash
# round 5, call C: the whole `-m gpu` suite at the round's kernels (exact apron, base-2 + pairs in K3, streaming pre-pass, fold opt-in), then the
# profile collection (bench line with both CPU baselines, kernel trace, PMC passes, issue model with per-kernel ISA mixes), then the 16-frame
# FREE-RUNNING comparison at configs[4]'s own 7680 x 4320 (RFX_TEST_8K=1)
O=gpurun_out/r05_c; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
PMC_TIMEOUT=150 timeout 900 bash tools/collect_profiles.sh r05_final > $O/collect.log 2>&1
tail -3 $O/collect.log
cat gpurun_out/r05_final/issue_model.txt | cut -c1-220
RFX_TEST_8K=1 timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "free_running and 7680" > $O/pytest_free_running_8k.log 2>&1
grep "frame \|K2 \|passed\|failed\|Error\|assert" $O/pytest_free_running_8k.log | cut -c1-330 | tail -60
