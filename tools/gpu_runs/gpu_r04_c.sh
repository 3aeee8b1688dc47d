#!/bin/bash
# round 4, call D: where K1 spends its time — ablations (no exact texel fetch; no cell lookup either; no march), per-lane loops, no SLP
mkdir -p gpurun_out/r04_d
cd "$GRAFT_REPO_ROOT"
( timeout 600 bash tools/time_variants.sh ) > gpurun_out/r04_d/variants.txt 2>&1
grep "==\|K1 ssgi\|^frame\|sha1" gpurun_out/r04_d/variants.txt
