#!/bin/bash
# round 6, call R (the round's final state): the whole `-m gpu` suite with its stage-wise reports, the free-running 16 frames at 7680x4320 (RFX_TEST_8K=1), the default bench line
O=gpurun_out/r06_r; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x -s --durations=12 ) > $O/pytest_gpu_final.log 2>&1
grep -E "passed|failed|error|^real" $O/pytest_gpu_final.log | tail -4
( time RFX_TEST_8K=1 timeout 2400 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x -s -k "free_running and 7680" ) > $O/pytest_free_running_8k.log 2>&1
grep -E "passed|failed|error|^real" $O/pytest_free_running_8k.log | tail -4
( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python -c "
import json
j = json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'kernel_ms', j['kernel_ms'], 'roofline', j['roofline']['kernel'], j['roofline']['frac'], 'traffic', j['roofline']['traffic'])
print('configs4', j['configs4_8k']['ms_per_frame'], j['configs4_8k']['dominant_kernel_by_frame_time'])"
