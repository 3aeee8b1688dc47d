#!/bin/bash
# round 4, call F: the second look (exact 16x16 cells) before an exact texel is fetched
mkdir -p gpurun_out/r04_f
cd "$GRAFT_REPO_ROOT"
( timeout 600 bash tools/time_variants.sh ) > gpurun_out/r04_f/variants.txt 2>&1
grep "==\|K1 ssgi\|^frame\|sha1" gpurun_out/r04_f/variants.txt
