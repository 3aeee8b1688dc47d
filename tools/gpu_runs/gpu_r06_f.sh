#!/bin/bash
# round 6, call F: K2's experiments against HEAD's K2 on one box — in-frame time (twice), then counters per variant (one --pmc set per pass, K2 alone):
#   k2_pairs = the taps' y-lerps and the five-tap combination on (r,g)/(b,a) float2 pairs;  k2_lds = velocity planes at their own aprons (36 672 B of LDS);
#   k2_w8 = k2_lds with the register allocator held to 8 waves per SIMD (a fourth workgroup per CU)
O=gpurun_out/r06_f; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
LIBS="realism-effects_amd/csrc/variants/librfx_k2_head.so realism-effects_amd/csrc/variants/librfx_k2_pairs.so realism-effects_amd/csrc/variants/librfx_k2_lds.so realism-effects_amd/csrc/variants/librfx_k2_w8.so"
for rep in 1 2; do
for so in $LIBS; do
  echo "== $so"; timeout 300 python tools/quick_time.py --lib $PWD/$so 3840 2160 20 | grep "frame\|K2\|temporal0"
done
done > $O/frames.txt 2>&1
cat $O/frames.txt
cd /tmp
for so in $LIBS; do
  n=$(basename $so .so); i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
             "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/$O/pmc_${n}_$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_time.py --lib $GRAFT_REPO_ROOT/$so 3840 2160 3 K2 > $GRAFT_REPO_ROOT/$O/pmc_${n}_$i.log 2>&1 || echo "set $i failed for $n"
  done
  echo "== $n"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/$O/pmc_${n}_* | grep k2_temporal | sed 's/void (anonymous namespace)::k2_temporal_reproject<0, 2, true, false, true>(K2Ar/K2 /'
done > $GRAFT_REPO_ROOT/$O/k2_counters.txt 2>&1
cat $GRAFT_REPO_ROOT/$O/k2_counters.txt
rm -rf $GRAFT_REPO_ROOT/$O/pmc_*
