#!/bin/bash
# round 4, call R: K3 / K2 tile staging with all of a thread's loads in flight before the first texel is transformed (RFX_K3_STAGE_AHEAD 3 / RFX_K2_STAGE_AHEAD 2)
# against one round trip per texel (= 1); same values, same stores: sha1 of B0 and of the composed frame must not move
mkdir -p gpurun_out/r04_r
cd "$GRAFT_REPO_ROOT"
( timeout 400 bash tools/time_variants.sh ) > gpurun_out/r04_r/variants.txt 2>&1
grep "==\|K2 \|K3 \|K4 \|^frame\|sha1" gpurun_out/r04_r/variants.txt
