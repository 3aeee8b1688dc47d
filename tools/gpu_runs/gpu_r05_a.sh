#!/bin/bash
# round 5, call A: (1) which TCC counters this rocprofv3 offers; (2) K3 variants at 4K — exact apron vs the wide one, base-2 logarithms, float2 pairs
# across the textures, 64x16 / 128x8 tiles; (3) K1's tile dealing (round-robin vs XCD groups vs static): time + FETCH_SIZE per launch;
# (4) FETCH_SIZE on known request counts (tools/microbench/gather_fetch); (5) the OPT-IN folded compose against the reference GLSL at 4K and on the 8K band
O=gpurun_out/r05_a; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( cd /tmp && rocprofv3 -L 2>&1 | grep -i -E "TCC_EA0?_(RD|WR)|TCC_EA0?_RDREQ|FETCH_SIZE|WRITE_SIZE|TCC_BUBBLE|TCC_REQ|MALL|TCC_HIT|TCC_MISS" | head -80 ) > $O/counters_tcc.txt 2>&1
# (2) + (3) timing
( timeout 900 bash tools/time_variants.sh ) > $O/variants.txt 2>&1
grep "==\|^K1 ssgi\|^K2\|^K3\|^K4\|^frame\|sha1" $O/variants.txt
# (3) FETCH_SIZE of K1 per dealing
for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_k1_ssgi_*.so; do
  n=$(basename $so .so)
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_$n -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_time.py --lib $GRAFT_REPO_ROOT/$so 3840 2160 4 K1 ) > $O/pmc_$n.log 2>&1
done
python tools/pmc_summary.py $O/pmc_librfx_* 2>&1 | grep -i "k1_ssgi_march\|k1_prepare" | grep -v "^$" > $O/k1_fetch_size.txt
cat $O/k1_fetch_size.txt
# (4) calibration
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_gather -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/bin/gather_fetch ) > $O/gather_fetch.txt 2>&1
python tools/pmc_summary.py $O/pmc_gather >> $O/gather_fetch.txt 2>&1
for extra in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $extra -d $GRAFT_REPO_ROOT/$O/pmc_gx -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/bin/gather_fetch ) > $O/gx.log 2>&1 && python tools/pmc_summary.py $O/pmc_gx >> $O/gather_fetch.txt 2>&1
  rm -rf $O/pmc_gx
done
grep -v "^$" $O/gather_fetch.txt | tail -30
rm -rf $O/pmc_librfx_* $O/pmc_gather
# (5) the folded compose against the reference GLSL (the default path's stages run alongside: the usual strict assertions)
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "configs[2] or rows" > $O/pytest_fold_vs_reference.log 2>&1
grep "folded\|passed\|failed\|Error" $O/pytest_fold_vs_reference.log | cut -c1-400
