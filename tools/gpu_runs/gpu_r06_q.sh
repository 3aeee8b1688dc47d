#!/bin/bash
# round 6, call Q: after the oracle's conditioning model learnt SampleGGXVNDF's tangent-frame term: the 16-frame stage-wise test without its pinned exception,
# the BASELINE configs stage-wise, smoke()
O=gpurun_out/r06_q; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x -s -k "16_frames_stagewise or configs[0] or configs[1]- or configs[2]" > $O/pytest_stagewise.log 2>&1
grep -E "passed|failed|error" $O/pytest_stagewise.log | tail -3
grep "f10 K1 ssgi\|f0 K1 ssgi\|f1 K1 ssgi" $O/pytest_stagewise.log | cut -c1-260 | head -12
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
