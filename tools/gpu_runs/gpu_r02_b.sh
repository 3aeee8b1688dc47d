#!/bin/bash
# round 2, GPU call B: the whole -m gpu suite (no -x), bench N=1, Node single-rank ring smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_b; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q -s --durations=15 2>&1 ) > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu.log | tail -15
grep -E "^f[0-9] K" $OUT/pytest_gpu.log > $OUT/parity_lines.txt
( time python bench.py --steps 20 --warmup 3 ) > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-1500
node -e "
const r=require('./realism-effects_amd/js');
const id=r.commUniqueId(); const t=new r.TiledRenderer(192,108,0,1,0,id);
t.afterComposePass(); t.commWait(); t.sync(); console.log('node single-rank ring ok', id.length, t.tileRows)
" > $OUT/node_ring.log 2>&1; cat $OUT/node_ring.log
