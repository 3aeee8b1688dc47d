#!/bin/bash
# round 4, call P: the whole -m gpu suite at the fold build (ABI 17), the bench line, then the profile collection (tools/collect_profiles.sh r04_final)
mkdir -p gpurun_out/r04_p
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/r04_p/pytest_gpu.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r04_p/pytest_gpu.log | tail -6
grep "frame .* composed outside\|open pixels\|fold at" gpurun_out/r04_p/pytest_gpu.log
PMC_TIMEOUT=120 timeout 900 bash tools/collect_profiles.sh r04_final > gpurun_out/r04_p/collect.log 2>&1
tail -3 gpurun_out/r04_p/collect.log | cut -c1-400
cat gpurun_out/r04_final/errors.txt 2>/dev/null
cat gpurun_out/r04_final/issue_model.txt 2>/dev/null
