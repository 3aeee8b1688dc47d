#!/bin/bash
# development helper: time every csrc/variants/librfx_*.so at 4K (run on the GPU box)
cd "$(dirname "$0")/.."
for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_*.so; do
  echo "== $so"
  timeout 300 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 ${1:-} | grep -v "^scene"
done
