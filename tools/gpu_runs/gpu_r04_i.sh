#!/bin/bash
# round 4, call I: K1's refinement packed one ray per lane (ds_permute / ds_bpermute) against the two-slot form
mkdir -p gpurun_out/r04_i
cd "$GRAFT_REPO_ROOT"
( timeout 300 bash tools/time_variants.sh ) > gpurun_out/r04_i/variants.txt 2>&1
grep "==\|K1 \|^frame\|sha1" gpurun_out/r04_i/variants.txt
