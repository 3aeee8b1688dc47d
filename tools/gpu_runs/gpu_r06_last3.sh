#!/bin/bash
# round 6, on the final tree: one more batch of every fuzzer with fresh seeds
set -x
O=gpurun_out/r06_last3; mkdir -p $O
export LP_NUM_THREADS=32 OMP_NUM_THREADS=32
timeout 900 python tools/fuzz_effects.py --device --n 1500 --seed 301 > $O/fuzz_effects_device_1500_seed301.txt 2>&1; tail -1 $O/fuzz_effects_device_1500_seed301.txt | cut -c1-300
timeout 900 python tools/fuzz_hostsim.py --device --n 800 --seed 302 > $O/fuzz_device_800_seed302.txt 2>&1; tail -1 $O/fuzz_device_800_seed302.txt
timeout 900 python tools/fuzz_variants_vs_reference_gl.py --device --n 900 --seed 303 > $O/fuzz_variants_device_vs_reference_gl_900_seed303.txt 2>&1; tail -1 $O/fuzz_variants_device_vs_reference_gl_900_seed303.txt | cut -c1-300
timeout 900 python tools/fuzz_vs_reference_gl.py --device --n 800 --seed 304 > $O/fuzz_device_vs_reference_gl_800_seed304.txt 2>&1; tail -1 $O/fuzz_device_vs_reference_gl_800_seed304.txt | cut -c1-300
timeout 900 python tools/fuzz_aux_vs_reference_gl.py --device --n 600 --seed 305 > $O/fuzz_aux_device_600_seed305.txt 2>&1; tail -1 $O/fuzz_aux_device_600_seed305.txt
