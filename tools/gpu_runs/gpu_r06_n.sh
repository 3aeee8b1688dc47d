#!/bin/bash
# round 6, call N: the kernels under LLVM's other AMDGPU scheduling strategies (-mllvm -amdgpu-sched-strategy=...: max-ilp, max-memory-clause, iterative-minreg /
# -maxocc / -ilp) against the default, same box; sha1 of every stage output (scheduling must not change a bit)
O=gpurun_out/r06_n; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_sched_*.so realism-effects_amd/csrc/librfx_hip.so; do
  echo "== $so"; timeout 300 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 | grep -v "^scene\|K1t\|K1s\|halo"
done > $O/sched_strategies.txt 2>&1
cat $O/sched_strategies.txt | cut -c1-120
