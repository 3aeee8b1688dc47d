#!/bin/bash
# round 6, call M: tools/collect_profiles.sh r06_final (bench line with both CPU baselines and the configs[4] extra, kernel trace without the solo loops, PMC passes, issue model), the other BASELINE configs
O=gpurun_out/r06_m; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
PMC_TIMEOUT=150 timeout 1500 bash tools/collect_profiles.sh r06_final > $O/collect.log 2>&1
tail -3 $O/collect.log | cut -c1-400
cat gpurun_out/r06_final/issue_model.txt | cut -c1-220
for cfg in "1920 1080 8 2 0 300" "1920 1080 20 5 1 300" "3840 2160 20 5 1 100" "7680 4320 40 5 3 16"; do timeout 300 python tools/run_config.py $cfg 2>&1 | tail -1; done > gpurun_out/r06_final/hip_configs.txt; cat gpurun_out/r06_final/hip_configs.txt
