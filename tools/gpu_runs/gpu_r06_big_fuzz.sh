#!/bin/bash
# round 6, the last call: tests/test_gpu_fuzz.py on the device, then large batches of the four fuzzers (they take seconds per hundred cases since
# the checkers' OpenMP teams are bounded).
set -x
O=gpurun_out/r06_big; mkdir -p $O
export LP_NUM_THREADS=32 OMP_NUM_THREADS=32
( time timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q ) > $O/pytest_gpu_fuzz.log 2>&1; tail -4 $O/pytest_gpu_fuzz.log
timeout 1500 python tools/fuzz_effects.py --device --n 2000 --seed 201 > $O/fuzz_effects_device_2000_seed201.txt 2>&1; tail -2 $O/fuzz_effects_device_2000_seed201.txt | cut -c1-400
timeout 1500 python tools/fuzz_hostsim.py --device --n 1000 --seed 202 > $O/fuzz_device_1000_seed202.txt 2>&1; tail -1 $O/fuzz_device_1000_seed202.txt
timeout 1500 python tools/fuzz_variants_vs_reference_gl.py --device --n 600 --seed 203 > $O/fuzz_variants_device_vs_reference_gl_600_seed203.txt 2>&1; tail -1 $O/fuzz_variants_device_vs_reference_gl_600_seed203.txt | cut -c1-300
timeout 1500 python tools/fuzz_variants_vs_reference_gl.py --device --n 200 --seed 204 --only-envmis > $O/fuzz_variants_device_vs_reference_gl_envmis_200_seed204.txt 2>&1; tail -1 $O/fuzz_variants_device_vs_reference_gl_envmis_200_seed204.txt | cut -c1-300
timeout 1500 python tools/fuzz_vs_reference_gl.py --device --n 400 --seed 205 > $O/fuzz_device_vs_reference_gl_400_seed205.txt 2>&1; tail -1 $O/fuzz_device_vs_reference_gl_400_seed205.txt | cut -c1-300
