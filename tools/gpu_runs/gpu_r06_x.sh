#!/bin/bash
# round 6, call X: the variants' differential fuzz in lock step (tools/fuzz_effects.py) against the product library on the device: 600 random
# option sets / sizes / cameras / environments through SSGIEffect, every draw on identical inputs against the C restatement; and 400 more
# cases of the default chain with row tilings (tools/fuzz_hostsim.py --device, another seed).
set -x
mkdir -p gpurun_out/r06_x
timeout 1500 python tools/fuzz_effects.py --device --n 600 --seed 7 > gpurun_out/r06_x/fuzz_effects_device_seed7.txt 2>&1
tail -6 gpurun_out/r06_x/fuzz_effects_device_seed7.txt | cut -c1-700
timeout 1500 python tools/fuzz_hostsim.py --device --n 400 --seed 8 > gpurun_out/r06_x/fuzz_device_seed8.txt 2>&1
tail -4 gpurun_out/r06_x/fuzz_device_seed8.txt | cut -c1-700
