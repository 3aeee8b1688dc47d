#!/bin/bash
# round 6, the remaining GPU minutes: every fuzzer once more on the final tree (after the cube chain for any face size), larger batches, fresh seeds
set -x
O=gpurun_out/r06_last4; mkdir -p $O
export LP_NUM_THREADS=32 OMP_NUM_THREADS=32
timeout 1200 python tools/fuzz_effects.py --device --n 3000 --seed 501 > $O/fuzz_effects_device_3000_seed501.txt 2>&1; tail -1 $O/fuzz_effects_device_3000_seed501.txt | cut -c1-300
timeout 1200 python tools/fuzz_hostsim.py --device --n 1500 --seed 502 > $O/fuzz_device_1500_seed502.txt 2>&1; tail -1 $O/fuzz_device_1500_seed502.txt
timeout 1200 python tools/fuzz_variants_vs_reference_gl.py --device --n 1800 --seed 503 > $O/fuzz_variants_device_vs_reference_gl_1800_seed503.txt 2>&1; tail -1 $O/fuzz_variants_device_vs_reference_gl_1800_seed503.txt | cut -c1-300
timeout 1200 python tools/fuzz_vs_reference_gl.py --device --n 1600 --seed 504 > $O/fuzz_device_vs_reference_gl_1600_seed504.txt 2>&1; tail -1 $O/fuzz_device_vs_reference_gl_1600_seed504.txt | cut -c1-300
timeout 600 python tools/fuzz_aux_vs_reference_gl.py --device --n 1000 --seed 505 > $O/fuzz_aux_device_1000_seed505.txt 2>&1; tail -1 $O/fuzz_aux_device_1000_seed505.txt
