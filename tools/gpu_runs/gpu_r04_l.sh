#!/bin/bash
# round 4, call L: the compose draw folded into the last denoise launch against two launches (A/B at 4K), the fold's tests on the device
mkdir -p gpurun_out/r04_l
cd "$GRAFT_REPO_ROOT"
( timeout 300 bash tools/time_variants.sh ) > gpurun_out/r04_l/variants.txt 2>&1
grep "==\|K3 pass1\|K4\|^frame\|sha1" gpurun_out/r04_l/variants.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "folded" > gpurun_out/r04_l/pytest_folded.log 2>&1
grep "fold at\|folded vs\|passed\|failed" gpurun_out/r04_l/pytest_folded.log
