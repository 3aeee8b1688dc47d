#!/bin/bash
# Round 3, GPU call G: same-box A/B of the round's last knobs (pre-pass stream on / off, K1 workgroup rows 2 / 4 / 8, K2 tile rows 4 / 8),
# and whether bench.py survives a counter pass again now that the in-tree library loads after torch (as in round 2).
set -x
O=gpurun_out/r03_g; mkdir -p $O
for rep in 1 2; do
  echo "== default (rep $rep)"; timeout 100 python tools/quick_time.py 3840 2160 20 2>&1 | grep -E "^K|^frame|sha1" | tee -a $O/ab.txt
  for so in realism-effects_amd/csrc/variants/librfx_*.so; do
    echo "== $so (rep $rep)" | tee -a $O/ab.txt
    timeout 100 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 2>&1 | grep -E "^K1 ssgi|^K2|^frame|ssgi sha1" | tee -a $O/ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$O/pmc -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stream-copy > $GRAFT_REPO_ROOT/$O/pmc.log 2>&1; echo "pmc(bench) rc=$?"
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/$O/pmcq -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_time.py 3840 2160 2 > $GRAFT_REPO_ROOT/$O/pmcq.log 2>&1; echo "pmc(quick_time) rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/$O/pmc $GRAFT_REPO_ROOT/$O/pmcq 2>/dev/null | grep -E "k1_ssgi|k2_temp|k3_tiled|k4_comp" | cut -c1-260
