#!/bin/bash
# round 4, call T: K3's decoded-geometry plane (pass 0 writes normal.xyz + roughness per texel, later passes stage from it instead of decoding the G-buffer texel)
# against every pass decoding what it stages (a_nogeom: the same library built with -DRFX_K3_GEOM=0), at 4K (one later pass) and with configs[4]'s options at 8K (five)
mkdir -p gpurun_out/r04_t
cd "$GRAFT_REPO_ROOT"
( timeout 400 bash tools/time_variants.sh ) > gpurun_out/r04_t/variants.txt 2>&1
grep "==\|K3 \|^frame\|sha1" gpurun_out/r04_t/variants.txt
( timeout 200 python tools/run_config.py 7680 4320 40 5 3 16 2>&1 | grep -v "^dump gen" ) > gpurun_out/r04_t/configs4_geom.txt 2>&1
cat gpurun_out/r04_t/configs4_geom.txt
