#!/bin/bash
# round 5, call H: does the next frame's trace fill what K2-K4 leave idle?  (tools/overlap_probe.py: two contexts / streams on one GPU)
O=gpurun_out/r05_h; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python tools/overlap_probe.py 3840 2160 50 > $O/overlap_probe_4k.txt 2>&1; tail -8 $O/overlap_probe_4k.txt
