#!/bin/bash
# round 6, call D: replay of K1's / K2's hot basic blocks (VALU only) with the compiler's registers and with a parity-balanced renaming
O=gpurun_out/r06_d; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
for b in replay_k1 replay_k2; do timeout 100 tools/microbench/bin/$b > $O/$b.txt 2>&1; cat $O/$b.txt; done
