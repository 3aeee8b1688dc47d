#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_h; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -q -s --durations=5 2>&1 ) > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR|^real" $OUT/pytest_gpu.log | tail -8
grep -E "UNEXPLAINED [1-9]" $OUT/pytest_gpu.log | head -5
( time timeout 420 python tools/parity_configs.py --impl hip --size 7680x4320 --steps 40 --refine 5 --it 3 --frames 3 --compare-from 2 --out $OUT/parity_configs4_frame2.txt ) 2>&1 | tee $OUT/parity_configs4_frame2.log | grep -E "^#|real|^f2 K(2|4)|^f2 K3 pass[15]" | grep -v "^# OpenGL" | cut -c1-200
ls -la $OUT
