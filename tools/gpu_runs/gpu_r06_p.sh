#!/bin/bash
# round 6, call P: K1 16-wave workgroups (in-tree) against 8-wave ones (variants/librfx_k1_w8.so), same box, IN-FRAME (quick_time's frame loop) at 1080p, 4K and a 270-row tile-sized frame
O=gpurun_out/r06_p; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for size in "1920 1080" "3840 2160" "3840 270"; do for rep in 1 2; do for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_k1_w8.so; do
  echo "== $size $so"; timeout 300 python tools/quick_time.py --lib $PWD/$so $size 40 | grep "K1 ssgi\|^frame\|K1+K2"
done; done; done > $O/k1_w16_vs_w8_in_frame.txt 2>&1
cat $O/k1_w16_vs_w8_in_frame.txt | cut -c1-110
