#!/bin/bash
# round 4, call U (last): the bench line, then the whole -m gpu suite, at the round's final commit
mkdir -p gpurun_out/r04_u
cd "$GRAFT_REPO_ROOT"
timeout 400 python bench.py > gpurun_out/r04_u/bench.json 2> gpurun_out/r04_u/bench.err
tail -1 gpurun_out/r04_u/bench.json | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r04_u/pytest_gpu.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r04_u/pytest_gpu.log | tail -4
