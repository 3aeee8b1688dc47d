#!/bin/bash
# round 4, call Q: the other BASELINE configs on the round's final kernels (tools/run_config.py), the bench line with the committed r04_final counters,
# and the WHOLE-frame 8K parity case (RFX_TEST_8K=1: three distinct 7680x4320 frames, every stage output compared and proven) re-run after the bound change
mkdir -p gpurun_out/r04_q
cd "$GRAFT_REPO_ROOT"
( for c in "1920 1080 8 2 0 300" "1920 1080 20 5 1 300" "3840 2160 20 5 1 200" "7680 4320 40 5 3 16"; do timeout 300 python tools/run_config.py $c 2>&1 | grep -v "^dump gen"; done ) > gpurun_out/r04_q/hip_configs.txt 2>&1
cat gpurun_out/r04_q/hip_configs.txt
timeout 600 python bench.py > gpurun_out/r04_q/bench.json 2> gpurun_out/r04_q/bench.err
tail -1 gpurun_out/r04_q/bench.json | cut -c1-400
RFX_TEST_8K=1 timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "8K" > gpurun_out/r04_q/configs4_8k_whole_frames.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r04_q/configs4_8k_whole_frames.txt | tail -5
grep -c "UNEXPLAINED 0" gpurun_out/r04_q/configs4_8k_whole_frames.txt
