#!/bin/bash
# Round 3, GPU call F (short): the depth pre-pass on its own stream — per-kernel times, the bench line, sha1s — and ONE counter pass with
# a tight timeout (call E's PMC passes hung for 600 s each: is it the torch stream-copy kernels under counter collection?).
set -x
O=gpurun_out/r03_f; mkdir -p $O
timeout 100 python tools/quick_time.py 3840 2160 20 > $O/quick_time_4k.txt 2>&1; cat $O/quick_time_4k.txt
timeout 150 python bench.py --no-cpu-baseline --checksum > $O/bench.json 2> $O/bench.err; cut -c1-700 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$O/pmc -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stream-copy > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1; echo "pmc rc=$?"
tail -5 $GRAFT_REPO_ROOT/$O/pmc.log
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/$O/pmc 2>/dev/null | grep -E "k1_ssgi|k2_temp|k3_tiled|k4_comp" | cut -c1-300
