#!/bin/bash
# round 4, call S: K2 neighbourhood AABB with immediate LDS offsets on tiles whose apron lies inside the frame (RFX_K2_INTERIOR) against computed clamped addresses; same texels
# against one round trip per texel (= 1); same values, same stores: sha1 of B0 and of the composed frame must not move
mkdir -p gpurun_out/r04_s
cd "$GRAFT_REPO_ROOT"
( timeout 400 bash tools/time_variants.sh ) > gpurun_out/r04_s/variants.txt 2>&1
grep "==\|K2 \|K3 \|K4 \|^frame\|sha1" gpurun_out/r04_s/variants.txt
