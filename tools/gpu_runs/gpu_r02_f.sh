#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_f; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -s --durations=6 2>&1 ) > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR|^real|s call" $OUT/pytest_gpu.log | tail -12
grep -E "UNEXPLAINED [1-9]" $OUT/pytest_gpu.log | head; grep -A3 "unexplained (y" $OUT/pytest_gpu.log | head -40
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.json | cut -c1-300
