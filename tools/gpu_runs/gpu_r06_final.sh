#!/bin/bash
# round 6, the final state after the peer mode in the Node host, the cleared last-error slot and K1's quad-partner correction (importance-sampling
# specialisation only: the default path's code objects are unchanged): the whole `-m gpu` suite with its stage-wise reports, the profile
# collection (bench line, rocprofv3 kernel trace, one --pmc set per pass) into gpurun_out/r06_final, the default bench line once more.
O=gpurun_out/r06_fin; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -m gpu -q -x -s --durations=12 ) > $O/pytest_gpu_final.log 2>&1
grep -E "passed|failed|error|^real" $O/pytest_gpu_final.log | tail -4
( time bash tools/collect_profiles.sh r06_final ) > $O/collect.log 2>&1
tail -3 $O/collect.log
cd "$GRAFT_REPO_ROOT"
( time python bench.py ) > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python -c "
import json
j = json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'kernel_ms', j['kernel_ms'], 'roofline', j['roofline']['kernel'], j['roofline']['frac'], 'traffic', j['roofline']['traffic'])
print('configs4', j['configs4_8k']['ms_per_frame'], j['configs4_8k']['dominant_kernel_by_frame_time'])"
