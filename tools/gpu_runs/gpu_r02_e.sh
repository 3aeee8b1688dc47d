#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_e; mkdir -p $OUT
export TMPDIR=/tmp
rm -f /tmp/multirank.err; timeout 600 bash tools/multirank_one_gpu.sh > $OUT/multirank.txt 2>&1; cat $OUT/multirank.txt; grep -B2 -A8 "Traceback" /tmp/multirank.err | head -30
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bench_multi_rank" 2>&1 | tail -3
python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "configs[1]" > $OUT/cfg1.log 2>&1; grep -E "UNEXPLAINED [1-9]|unexplained|impl  |ref   |oracle|passed|failed" $OUT/cfg1.log | head -60
python __graft_entry__.py smoke 2>&1 | tail -4
