#!/bin/bash
# round 6, call A: K2's LDS diet (velocity planes at their own aprons: 45 696 -> 36 672 B) alone and with the register allocator held to 8 waves per SIMD
# (a fourth workgroup per CU), against HEAD's K2 — same box, same call; sha1 of TEMPORAL0 must agree
O=gpurun_out/r06_a; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/time_variants.sh "K2" > $O/k2_variants.txt 2>&1
cat $O/k2_variants.txt | grep -v "^K1+K2\|halo"
for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_k2_head.so realism-effects_amd/csrc/variants/librfx_k2_w8.so; do
  echo "== $so"; timeout 300 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 | grep "frame\|K2\|sha1"
done > $O/frames.txt 2>&1
cat $O/frames.txt
