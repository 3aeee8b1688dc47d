#!/bin/bash
# round 6, call S: K1 with its (min, max) cells as fp32 values truncated to 16 bits (a shift and a mask instead of two v_cvt_f32_f16 per cell; looser bounds) against half cells
O=gpurun_out/r06_s; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_k1_bf16.so; do
  echo "== $so"; timeout 300 python tools/quick_time.py --lib $PWD/$so 3840 2160 30 | grep "K1 ssgi\|^frame\|ssgi sha1"
done; done > $O/k1_bf16_cells.txt 2>&1
cat $O/k1_bf16_cells.txt | cut -c1-110
