#!/bin/bash
# round 4, call E: no-SLP build, per-lane loops; merged exact-texel fetches on / off, wave loops, ablations again
mkdir -p gpurun_out/r04_e
cd "$GRAFT_REPO_ROOT"
( timeout 600 bash tools/time_variants.sh ) > gpurun_out/r04_e/variants.txt 2>&1
grep "==\|K1 ssgi\|^frame\|sha1" gpurun_out/r04_e/variants.txt
