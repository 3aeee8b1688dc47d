#!/bin/bash
# round 4, call K (first call of the resumed session): the bench line at HEAD + the kernel trace of the bench command
mkdir -p gpurun_out/r04_k
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > gpurun_out/r04_k/bench.json 2> gpurun_out/r04_k/bench.err
tail -1 gpurun_out/r04_k/bench.json | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04_k/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --spinup 50 --no-cpu-baseline --no-stream-copy --no-cold > $GRAFT_REPO_ROOT/gpurun_out/r04_k/trace.log 2>&1
cp $(find $GRAFT_REPO_ROOT/gpurun_out/r04_k/trace -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_k/kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r04_k/trace
head -12 $GRAFT_REPO_ROOT/gpurun_out/r04_k/kernel_stats.csv
