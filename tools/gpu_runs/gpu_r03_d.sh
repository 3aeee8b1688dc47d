#!/bin/bash
# Round 3, GPU call D: the 8K band case after the exact bicubic quotients in K2, the new tests (hit rows, smoke fallback), what the bounded
# history gather would move at N = 2 / 4 / 8, per-kernel times.
set -x
O=gpurun_out/r03_d; mkdir -p $O
timeout 120 python tools/quick_time.py 3840 2160 20 > $O/quick_time_4k.txt 2>&1; cat $O/quick_time_4k.txt
timeout 200 python tools/history_rows_report.py 3840x2160 2 > $O/history_rows_4k.txt 2>&1; cat $O/history_rows_4k.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hit_rows or smoke_fallback or single_rank_ring" > $O/new_tests.txt 2>&1; tail -3 $O/new_tests.txt
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x -s -k "8K and rows" > $O/configs4_8k_band.txt 2>&1
grep -E "passed|failed|UNEXPLAINED [1-9]|K2 temporal" $O/configs4_8k_band.txt | tail -12
