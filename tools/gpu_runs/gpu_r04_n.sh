#!/bin/bash
# round 4, call N: K1 at 8 / 6 / 4 / 2 wavefronts per SIMD (persistent grid size): how much of the march is latency hidden by other waves
mkdir -p gpurun_out/r04_n
cd "$GRAFT_REPO_ROOT"
( timeout 400 bash tools/time_variants.sh K1 ) > gpurun_out/r04_n/variants.txt 2>&1
grep "==\|K1 \|K1t\|K4\|^frame\|sha1" gpurun_out/r04_n/variants.txt
cd /tmp && export TMPDIR=/tmp
