#!/bin/bash
# round 5, call I (the round's final kernels: K1 hit tests in the exec region + power-of-two table pitch, K3 pass 0 corner shave): the whole `-m gpu` suite with
# its stage-wise reports (-s), the profile collection (bench line with both CPU baselines, kernel trace, PMC passes, issue model), the other BASELINE configs
O=gpurun_out/r05_i; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -s > $O/pytest_gpu_final.log 2>&1
grep "passed\|failed\|error" $O/pytest_gpu_final.log | tail -3
PMC_TIMEOUT=120 timeout 700 bash tools/collect_profiles.sh r05_final > $O/collect.log 2>&1
tail -2 $O/collect.log | cut -c1-400
cat gpurun_out/r05_final/issue_model.txt | cut -c1-220
for cfg in "1920 1080 8 2 0 300" "1920 1080 20 5 1 300" "3840 2160 20 5 1 100" "7680 4320 40 5 3 16"; do timeout 300 python tools/run_config.py $cfg 2>&1 | tail -1; done > $O/hip_configs.txt; cat $O/hip_configs.txt
