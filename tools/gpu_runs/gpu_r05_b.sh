#!/bin/bash
# round 5, call B: (1) variants at 4K: K1's pair-form march step and streaming pre-pass against the round-4 forms, K2's DPP neighbourhood AABB;
# (2) K1's FETCH_SIZE against occupancy and tile height; (3) the gather microbenchmark with kernel durations; (4) the bench line (in-frame kernel times);
# (5) stage-wise parity of configs[1], configs[2] with the new K3 arithmetic (base-2 logarithms, pairs) + the folded stage; (6) the free-running table
O=gpurun_out/r05_b; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 600 bash tools/time_variants.sh ) > $O/variants.txt 2>&1
grep "==\|^K1 ssgi\|^K2\|^K3\|^K4\|^frame\|sha1" $O/variants.txt
for so in realism-effects_amd/csrc/librfx_hip.so realism-effects_amd/csrc/variants/librfx_k1_ssgi_occ4.so realism-effects_amd/csrc/variants/librfx_k1_ssgi_occ6.so realism-effects_amd/csrc/variants/librfx_k1_ssgi_th4.so; do
  n=$(basename $so .so)
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_$n -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_time.py --lib $GRAFT_REPO_ROOT/$so 3840 2160 4 K1 ) > $O/pmc_$n.log 2>&1
  echo "== $n" >> $O/k1_fetch_size.txt
  python tools/pmc_summary.py $O/pmc_$n 2>&1 | grep -i "k1_ssgi_march\|k1_prepare" >> $O/k1_fetch_size.txt
done
cat $O/k1_fetch_size.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/gather_trace -o g --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/bin/gather_fetch ) > $O/gather_trace.log 2>&1
cp $(find $O/gather_trace -name "*kernel_stats.csv" | head -1) $O/gather_kernel_stats.csv 2>/dev/null; cat $O/gather_kernel_stats.csv | cut -c1-200
rm -rf $O/pmc_librfx_* $O/gather_trace
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('value','ms_per_step','ms_per_step_cold','kernel_ms','k1_prepass_ms','kernel_ms_solo','ms_per_step_compose_fold_opt_in'): print(k, d.get(k))
print(d['roofline'])"
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "configs[1]-1920 or configs[2]" > $O/pytest_stagewise.log 2>&1
grep "K3\|K4\|passed\|failed\|Error" $O/pytest_stagewise.log | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "free_running" > $O/pytest_free_running.log 2>&1
grep "frame \|K2 \|passed\|failed\|Error" $O/pytest_free_running.log | cut -c1-330
