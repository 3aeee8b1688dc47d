#!/bin/bash
# Round 3, final call 3: the parity reports of BASELINE configs[0], [1], [2] and configs[4]'s options with the FINAL kernels (BASELINE.md's
# table), the re-worked configs[0] end-to-end test, K1's sha1 after the exact constant fold.
set -x
O=gpurun_out/r03_final3; mkdir -p $O
timeout 100 python tools/quick_time.py 3840 2160 20 2>&1 | grep -E "^frame|ssgi sha1|^K" | tee $O/quick_time_4k.txt   # K1 sha1: ffbc919111870b36 before the fold
timeout 200 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -m gpu -q -x -s -k "config0_through or single_rank_ring" > $O/tests.txt 2>&1; grep -E "effect K2|K1 texels|passed|failed" $O/tests.txt | tail -10
timeout 120 python tools/parity_configs.py --impl hip --size 1920x1080 --steps 8 --refine 2 --it 0 --frames 2 --out $O/parity_configs0.txt > /dev/null 2>&1; tail -9 $O/parity_configs0.txt
timeout 120 python tools/parity_configs.py --impl hip --size 1920x1080 --steps 20 --it 1 --frames 2 --out $O/parity_configs1.txt > /dev/null 2>&1; tail -9 $O/parity_configs1.txt
timeout 300 python tools/parity_configs.py --impl hip --size 3840x2160 --steps 20 --it 1 --frames 2 --out $O/parity_configs2.txt > /dev/null 2>&1; tail -9 $O/parity_configs2.txt
timeout 200 python tools/parity_configs.py --impl hip --size 1920x1080 --steps 40 --it 3 --frames 3 --out $O/parity_configs4_options_1080p.txt > /dev/null 2>&1; tail -9 $O/parity_configs4_options_1080p.txt
