#!/bin/bash
# round 5, call G: K3's later passes with a geometry window smaller than the input window (40 176 B of LDS at 4K: four workgroups per CU instead of
# three) and their registers bounded to 64 (in-tree), against the same without the register bound (waves1: 70 VGPRs, three workgroups per CU) and
# the one-window kernels of the previous commit (onewindow).  Same sha1 expected everywhere.
O=gpurun_out/r05_g; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 400 bash tools/time_variants.sh ) > $O/variants.txt 2>&1
grep "==\|^K1 ssgi\|^K3\|^frame\|sha1" $O/variants.txt
( timeout 200 python tools/run_config.py 7680 4320 40 5 3 16 2>&1 | tail -1 ) > $O/hip_config4.txt; cat $O/hip_config4.txt
