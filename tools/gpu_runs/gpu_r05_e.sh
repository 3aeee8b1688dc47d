#!/bin/bash
# round 5, call E (session 2; call D's outputs were lost with the container): (1) LDS residency probe (how many K3 workgroups a CU holds at their LDS sizes);
# (2) K3 variants at 4K: interleaved later-pass staging off, the tap loop unrolled (pass 0 / later passes / both; 2, 4, 8 taps), 64x7 tiles;
# (3) the bench line with the committed counters of r05_final (roofline.traffic, issue_model); (4) the other BASELINE configs through SSGIEffect
O=gpurun_out/r05_e; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 60 tools/microbench/bin/lds_occupancy > $O/lds_occupancy.txt 2>&1; grep -v "^  dynamic" $O/lds_occupancy.txt | head -20
( timeout 420 bash tools/time_variants.sh K3 ) > $O/variants_k3.txt 2>&1
grep "==\|^K3\|sha1" $O/variants_k3.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-600
for cfg in "1920 1080 8 2 0 300" "1920 1080 20 5 1 300" "3840 2160 20 5 1 100" "7680 4320 40 5 3 16"; do timeout 300 python tools/run_config.py $cfg 2>&1 | tail -1; done > $O/hip_configs.txt; cat $O/hip_configs.txt
