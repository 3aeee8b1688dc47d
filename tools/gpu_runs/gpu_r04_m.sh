#!/bin/bash
# round 4, call M: K1's in-wave march compaction (one ray per lane once a wavefront's live rays fit) against the two-slot march, A/B at 4K + sha1;
# which kernels run while quick_time times "K4 compose" (0.130 ms in call L's folding build against 0.096)
mkdir -p gpurun_out/r04_m
cd "$GRAFT_REPO_ROOT"
( timeout 400 bash tools/time_variants.sh K1 ) > gpurun_out/r04_m/variants.txt 2>&1
grep "==\|K1 \|K1t\|K4\|^frame\|sha1" gpurun_out/r04_m/variants.txt
cd /tmp && export TMPDIR=/tmp
