#!/bin/bash
# GPU box: bench.py's N>1 flow with 4 and 3 ranks sharing the one GPU over gloo (RFX_BENCH_ONE_GPU=1) — middle ranks with two
# neighbours, an odd rank count — and the sha1 of the whole composed GI against the single-rank run of the same frame.
export RFX_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 4 --width 960 --height 540 --steps 2 --warmup 1 --no-cpu-baseline --checksum --no-extras 2>>/tmp/multirank.err | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4 ranks', j['config']['frame'], j['config']['tile_rows'], j['config']['halo_rows'], j['halo_violations'], j['compose_sha1'], j['ms_per_step'])
open('/tmp/frame.txt','w').write(j['config']['frame'])"
F=$(cat /tmp/frame.txt); W=${F%x*}; H=${F#*x}
python bench.py --width $W --height $H --steps 2 --warmup 1 --no-cpu-baseline --checksum --no-extras 2>>/tmp/multirank.err | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 rank ', j['config']['frame'], j['compose_sha1'], j['ms_per_step'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 3 --width 960 --height 540 --steps 2 --warmup 1 --no-cpu-baseline --checksum --no-extras 2>>/tmp/multirank.err | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('3 ranks', j['config']['frame'], j['config']['tile_rows'], j['halo_violations'], j['compose_sha1'])
open('/tmp/frame.txt','w').write(j['config']['frame'])"
F=$(cat /tmp/frame.txt); W=${F%x*}; H=${F#*x}
python bench.py --width $W --height $H --steps 2 --warmup 1 --no-cpu-baseline --checksum --no-extras 2>>/tmp/multirank.err | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 rank ', j['config']['frame'], j['compose_sha1'])"
