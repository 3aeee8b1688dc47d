#!/bin/bash
# round 5, call D: (1) K3's interleaved later-pass staging against the separate arrays (bit-identical), K1's grid from the occupancy API; (2) the bench line
# with the committed counters of r05_final (roofline.traffic, issue_model); (3) the other BASELINE configs through SSGIEffect; (4) the 16-frame stage-wise
# test with its log (which pixel is the documented "open" one); (5) the PCIe-inclusive rate
O=gpurun_out/r05_d; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 600 bash tools/time_variants.sh ) > $O/variants.txt 2>&1
grep "==\|^K1 ssgi\|^K2\|^K3\|^K4\|^frame\|sha1" $O/variants.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
for cfg in "1920 1080 8 2 0 300" "1920 1080 20 5 1 300" "3840 2160 20 5 1 100" "7680 4320 40 5 3 16"; do timeout 600 python tools/run_config.py $cfg 2>&1 | tail -1; done > $O/hip_configs.txt; cat $O/hip_configs.txt
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "stagewise_at_ages" > $O/pytest_16_frames_stagewise.log 2>&1
grep "open pixels\|unexplained (y\|UNEXPLAINED [1-9]\|passed\|failed" $O/pytest_16_frames_stagewise.log | cut -c1-300
