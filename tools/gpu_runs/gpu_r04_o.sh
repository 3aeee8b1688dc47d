#!/bin/bash
# round 4, call O: K1 march steps in pairs (k1_march_step2: step B speculated while step A's fetches are in flight) against one step per iteration
mkdir -p gpurun_out/r04_o
cd "$GRAFT_REPO_ROOT"
( timeout 400 bash tools/time_variants.sh ) > gpurun_out/r04_o/variants.txt 2>&1
grep "==\|K1 \|K1t\|^frame\|sha1" gpurun_out/r04_o/variants.txt
cd /tmp && export TMPDIR=/tmp
