#!/bin/bash
# development helper (GPU box): list the PMC counters and collect a few candidate sets over tools/quick_time.py
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_probe${PMC_TAG}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$PMC_LIST" ] && rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_time.py 3840 2160 2 ${PMC_STAGES} > $OUT/p$i.log 2>&1 || echo "set $i failed: $set"
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/p* 2>&1 | grep -v "^$"
