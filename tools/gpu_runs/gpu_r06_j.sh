#!/bin/bash
# round 6, call J: the peer-load history gather at BASELINE configs[3]'s size (3840x2160) with 2 / 4 / 8 processes sharing this one GPU: checksum against the
# single-rank run, and what each rank pulls per frame (history_exchange) — the bytes are real, the transfer rate is this device's own HBM, not xGMI
O=gpurun_out/r06_j; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp RFX_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1
C="--width 3840 --height 2160 --steps 8 --warmup 2 --no-cpu-baseline --checksum --no-extras --spinup 0"
python bench.py $C 2>$O/err1.txt | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 rank ', j['config']['frame'], j['compose_sha1'], j['ms_per_step'])" > $O/peer_4k.txt
for n in 2 4 8; do for mode in all peer; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --history-gather $mode $C 2>>$O/err$n.txt | grep "^{" | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n ranks $mode', j['config']['tile_rows'], j['config']['halo_rows'], j['halo_violations'], j['compose_sha1'], j['ms_per_step'], j['config']['history_exchange'])" >> $O/peer_4k.txt
done; done
cat $O/peer_4k.txt; tail -3 $O/err8.txt | cut -c1-300
