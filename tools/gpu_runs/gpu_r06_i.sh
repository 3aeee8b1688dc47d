#!/bin/bash
# round 6, call I: the device-driven history gather (rfx_peer_*: peer loads through HIP IPC mappings, flag barriers in mapped memory) — N processes on this
# one GPU, bit-identical to the single-rank run (2 ranks all-gather, 2 ranks peer, 3 ranks peer with a ragged tile)
O=gpurun_out/r06_i; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "multi_rank_flow" > $O/pytest_peer.log 2>&1; grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" $O/pytest_peer.log | tail -30
