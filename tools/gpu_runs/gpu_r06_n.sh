#!/bin/bash
# round 6, call N: the kernels under LLVM's other AMDGPU scheduling strategies (-mllvm -amdgpu-sched-strategy=...: max-ilp, max-memory-clause, iterative-minreg /
# -maxocc / -ilp) against the default, same box; sha1 of every stage output (scheduling must not change a bit)
O=gpurun_out/r06_n; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_sched_*.so realism-effects_amd/csrc/librfx_hip.so; do
  echo "== $so"; timeout 300 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 | grep -v "^scene\|K1t\|K1s\|halo"
done > $O/sched_strategies.txt 2>&1
cat $O/sched_strategies.txt | cut -c1-120
# ... and K1 with 16-wave workgroups (two per CU: the table copy and its barrier paid half as often), at 4K and at 1080p
for size in "3840 2160" "1920 1080"; do for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_k1_w16.so realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_k1_w16.so; do
  echo "== $size $so"; timeout 300 python tools/quick_time.py --lib $PWD/$so $size 20 K1 | grep "K1 ssgi\|ssgi sha1"
done; done > $O/k1_16_wave_workgroups.txt 2>&1
cat $O/k1_16_wave_workgroups.txt | cut -c1-120
