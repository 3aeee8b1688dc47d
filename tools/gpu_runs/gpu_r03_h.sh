#!/bin/bash
# round 3, call H: the workgroup-compacting K1 march (build knob RFX_K1_COMPACT, csrc/build_variants.sh) against the pixel-per-lane march —
# same texels (sha1) and the time of the fused launch, the trace and the whole frame at 4K; then 1080p 8/2 and 8K 40/5 geometry for the winner.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03_h
OUT=gpurun_out/r03_h/variants.txt
: > $OUT
for so in realism-effects_amd/csrc/variants/librfx_k1_ssgi_*.so; do
  echo "== $so" >> $OUT
  timeout 150 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 ${ONLY:-} >> $OUT 2>&1 || echo "FAILED rc=$?" >> $OUT
done
grep "^==\|^K1 ssgi\|^K2\|^K3\|^K4\|^frame\|sha1\|FAILED" $OUT
