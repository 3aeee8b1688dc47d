#!/bin/bash
# Round 3, final call 1: the whole default -m gpu suite (incl. the 8K band case) on the final build, smoke, the other configs' rates.
set -x
O=gpurun_out/r03_final1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^[0-9.]+s call" $O/pytest_gpu.log | tail -16
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for cfg in "1920 1080 8 2 0 8" "1920 1080 20 5 1 8" "7680 4320 40 5 3 16"; do timeout 200 python tools/run_config.py $cfg 2>&1 | tail -2 | tee -a $O/hip_configs.txt; done
