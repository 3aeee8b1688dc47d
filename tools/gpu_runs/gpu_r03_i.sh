#!/bin/bash
# round 3, call I: where does bench.py stop under `rocprofv3 --pmc` (it answers under --kernel-trace; tools/quick_time.py answers under --pmc)?
# faulthandler prints every Python thread's stack after 45 s; second run with the depth pre-pass on the draw stream (RFX_K1_PREP_STREAM=0).
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r03_i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WRAP="import faulthandler, sys, runpy; faulthandler.dump_traceback_later(45, exit=True); sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-stream-copy']; runpy.run_path('$ROOT/bench.py', run_name='__main__')"
timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/a -o p --output-format csv -- python -c "$WRAP" > $OUT/a.log 2>&1; echo "A rc=$?" >> $OUT/a.log
RFX_K1_PREP_STREAM=0 timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/b -o p --output-format csv -- python -c "$WRAP" > $OUT/b.log 2>&1; echo "B rc=$?" >> $OUT/b.log
rm -rf $OUT/a $OUT/b
tail -40 $OUT/a.log; echo ----; tail -15 $OUT/b.log
