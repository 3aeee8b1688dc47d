#!/bin/bash
# round 6, the last full run: the whole `-m gpu` suite on the final tree (with tests/test_gpu_fuzz.py), then smoke() and the default bench line
O=gpurun_out/r06_fin2; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -q -x -s --durations=12 ) > $O/pytest_gpu_final.log 2>&1
grep -E "passed|failed|error|^real" $O/pytest_gpu_final.log | tail -4
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err; python -c "
import json
j = json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'ms_per_step', j['ms_per_step'], 'roofline', j['roofline']['frac'])"
