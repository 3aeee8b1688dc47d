#!/bin/bash
# round 6, call E: K2 with the perspective inverse projection's zeros dropped (k2_ss_to_ws) against HEAD's K2 — same box; sha1s must agree
O=gpurun_out/r06_e; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_*.so; do
  echo "== $so"; timeout 300 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 | grep "frame\|K2\|sha1"
done
done > $O/frames.txt 2>&1
cat $O/frames.txt
