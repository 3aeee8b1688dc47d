#!/bin/bash
# round 6, call Y: the kernels against the REFERENCE'S OWN GLSL on llvmpipe at random sizes / step counts / option values, strict metric with
# proofs (tools/fuzz_vs_reference_gl.py --device: 300 cases), and the lock-step variants fuzz with TRAAEffect cases (200 cases).
set -x
mkdir -p gpurun_out/r06_y
export LP_NUM_THREADS=32
timeout 1700 python tools/fuzz_vs_reference_gl.py --device --n 300 --seed 12 > gpurun_out/r06_y/fuzz_device_vs_reference_gl_seed12.txt 2>&1
tail -5 gpurun_out/r06_y/fuzz_device_vs_reference_gl_seed12.txt | cut -c1-700
timeout 900 python tools/fuzz_effects.py --device --n 200 --seed 13 > gpurun_out/r06_y/fuzz_effects_device_seed13.txt 2>&1
tail -3 gpurun_out/r06_y/fuzz_effects_device_seed13.txt | cut -c1-700
