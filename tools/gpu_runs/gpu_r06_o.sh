#!/bin/bash
# round 6, call O: K1 with 16-wave workgroups shipped (no environment map): sha1s, the four BASELINE configs through run_config.py, the quick GPU subset
O=gpurun_out/r06_o; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python tools/quick_time.py 3840 2160 20 2>&1 | grep -v "^scene" > $O/quick_time_4k.txt; cat $O/quick_time_4k.txt
for cfg in "1920 1080 8 2 0 300" "1920 1080 20 5 1 300" "3840 2160 20 5 1 100" "7680 4320 40 5 3 16"; do timeout 300 python tools/run_config.py $cfg 2>&1 | tail -1; done > $O/hip_configs.txt; cat $O/hip_configs.txt
timeout 900 python -m pytest tests -m "gpu and quick" -q -x > $O/pytest_quick.log 2>&1; grep -E "passed|failed|error" $O/pytest_quick.log | tail -3
