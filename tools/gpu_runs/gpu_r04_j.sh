#!/bin/bash
# round 4, call J: the whole -m gpu suite at the packed-refinement build (strict-metric twins, 16-frame sequence, 4K streamed test), then the bench line
mkdir -p gpurun_out/r04_j
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/r04_j/pytest_gpu.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r04_j/pytest_gpu.log | tail -6
grep "frame .* composed outside\|open pixels" gpurun_out/r04_j/pytest_gpu.log
grep -A42 "test_zz_measured" gpurun_out/r04_j/pytest_gpu.log | grep " %$" | head -45
timeout 600 python bench.py > gpurun_out/r04_j/bench.json 2> gpurun_out/r04_j/bench.err
tail -1 gpurun_out/r04_j/bench.json | cut -c1-600
