#!/bin/bash
# round 6, call W: the differential fuzz (tools/fuzz_hostsim.py) against the product library on the device: 200 random cases of odd frame
# sizes, options and row tilings, every stage against the C restatement on identical inputs.
set -x
mkdir -p gpurun_out/r06_w
timeout 1500 python tools/fuzz_hostsim.py --device --n 200 --seed 6 > gpurun_out/r06_w/fuzz_device_seed6.txt 2>&1
tail -25 gpurun_out/r06_w/fuzz_device_seed6.txt
