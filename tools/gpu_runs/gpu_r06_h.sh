#!/bin/bash
# round 6, call H: after the fold removal + K1/K3 knob clean-up + K2 pairs: the sha1s of every 4K stage output (must equal round 5's: 02080d3d / 221985d2 / c28a2b8e /
# c5210c1c / cc4f1bea), the frame time, and the quick parity subset
O=gpurun_out/r06_h; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python tools/quick_time.py 3840 2160 20 > $O/quick_time_4k.txt 2>&1; cat $O/quick_time_4k.txt | grep -v "^scene"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_abi_and_host.py -m gpu -q -x > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
