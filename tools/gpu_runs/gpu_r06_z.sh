#!/bin/bash
# round 6, call Z: after the quad-partner correction in K1's importance-sampling specialisation (a partner outside an odd-sized target runs the
# fragment): the environment tests, the lock-step fuzz against the restatement (300 cases) and THE KERNELS AGAINST THE REFERENCE CHAIN over the
# variants (tools/fuzz_variants_vs_reference_gl.py --device, prebuilt programs: 120 cases + 40 odd-sized importance-sampling cases).
set -x
mkdir -p gpurun_out/r06_z
export LP_NUM_THREADS=32 OMP_NUM_THREADS=32
timeout 600 python -m pytest tests -q -m gpu -k "env_map or importance or cube" 2>&1 | tail -4 > gpurun_out/r06_z/pytest_env.log; cat gpurun_out/r06_z/pytest_env.log
timeout 900 python tools/fuzz_effects.py --device --n 300 --seed 17 > gpurun_out/r06_z/fuzz_effects_device_seed17.txt 2>&1
tail -2 gpurun_out/r06_z/fuzz_effects_device_seed17.txt | cut -c1-600
timeout 1200 python tools/fuzz_variants_vs_reference_gl.py --device --n 120 --seed 18 > gpurun_out/r06_z/fuzz_variants_device_vs_gl_seed18.txt 2>&1
tail -2 gpurun_out/r06_z/fuzz_variants_device_vs_gl_seed18.txt | cut -c1-600
timeout 600 python tools/fuzz_variants_vs_reference_gl.py --device --n 40 --seed 19 --only-envmis > gpurun_out/r06_z/fuzz_variants_device_vs_gl_envmis_seed19.txt 2>&1
tail -2 gpurun_out/r06_z/fuzz_variants_device_vs_gl_envmis_seed19.txt | cut -c1-600
