#!/bin/bash
# round 6, call L: the whole `-m gpu` suite with its stage-wise reports (-s) and durations, then the RFX_TEST_8K=1 cases (configs[4] whole 8K frames stage-wise on this
# round's kernels; the free-running 16 frames at 7680x4320)
O=gpurun_out/r06_l; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x -s --durations=25 ) > $O/pytest_gpu_full.log 2>&1
grep -E "passed|failed|error|^real" $O/pytest_gpu_full.log | tail -4
( time RFX_TEST_8K=1 timeout 2400 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x -s -k "8K" ) > $O/pytest_8k.log 2>&1
grep -E "passed|failed|error|^real" $O/pytest_8k.log | tail -4
