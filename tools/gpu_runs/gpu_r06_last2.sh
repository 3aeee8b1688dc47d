#!/bin/bash
# round 6: the variants fuzz with TRAAEffect cases and the cube / packer fuzz on the device, tests/test_gpu_fuzz.py as it is now
set -x
O=gpurun_out/r06_last2; mkdir -p $O
export LP_NUM_THREADS=32 OMP_NUM_THREADS=32
timeout 1500 python tools/fuzz_variants_vs_reference_gl.py --device --n 600 --seed 207 > $O/fuzz_variants_with_traa_device_vs_reference_gl_600_seed207.txt 2>&1; tail -1 $O/fuzz_variants_with_traa_device_vs_reference_gl_600_seed207.txt | cut -c1-300
timeout 900 python tools/fuzz_aux_vs_reference_gl.py --device --n 400 --seed 208 > $O/fuzz_aux_device_400_seed208.txt 2>&1; tail -1 $O/fuzz_aux_device_400_seed208.txt
( time timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q ) > $O/pytest_gpu_fuzz.log 2>&1; tail -5 $O/pytest_gpu_fuzz.log
