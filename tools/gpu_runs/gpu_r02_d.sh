#!/bin/bash
# round 2, GPU call D: multirank flow (stderr kept), full -m gpu suite, profile collection r02_final, other configs' rates
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_d; mkdir -p $OUT
export TMPDIR=/tmp
export RFX_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --width 960 --height 540 --steps 2 --warmup 1 --no-cpu-baseline --checksum --no-extras > $OUT/flow2.out 2> $OUT/flow2.err
grep -E "^\{" $OUT/flow2.out | cut -c1-400; grep -B2 -A12 "Traceback" $OUT/flow2.err | head -60
unset RFX_BENCH_ONE_GPU
( time timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 2>&1 ) > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR|^real|s call" $OUT/pytest_gpu.log | tail -16
grep -E "^f[0-9] K.*UNEXPLAINED [1-9]" $OUT/pytest_gpu.log | head
bash tools/collect_profiles.sh r02_final > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log | cut -c1-400
grep -E "asm|pk_|med3|cvt" gpurun_out/r02_final/valu_rates.txt | grep "8 waves"
for cfg in "1920 1080 8 2 0 16" "1920 1080 20 5 1 16" "7680 4320 40 5 3 16"; do python tools/run_config.py $cfg 2>&1 | tail -1; done > $OUT/hip_configs.txt; cat $OUT/hip_configs.txt
for cfg in "1920 1080 8 2 0 5" "1920 1080 20 5 1 5" "7680 4320 40 5 3 3"; do python tools/cpu_llvmpipe_rate.py $cfg 2>&1 | tail -1; done > $OUT/cpu_configs.txt; cat $OUT/cpu_configs.txt
