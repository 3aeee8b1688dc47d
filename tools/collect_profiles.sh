#!/bin/bash
# Run ON THE GPU BOX (through gpurun): collects the artefacts profiles/README.md describes into gpurun_out/<name>/, to be copied into
# profiles/<name>/ afterwards.   tools/collect_profiles.sh r01_final
set -u
NAME=${1:-profile}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$NAME
mkdir -p $OUT
python - "$OUT" <<'PY'
import json, subprocess, sys, time, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
try:
    commit = open(os.path.join(root, "gpurun_commit.txt")).read().strip()
except Exception:
    commit = "unknown"
json.dump({"profile": os.path.basename(sys.argv[1]), "git_commit": commit, "collected": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()),
           "command": "tools/collect_profiles.sh (3840x2160, steps 20/5, denoiseIterations 1; rocprofv3 --kernel-trace --stats of bench.py; one --pmc set per pass over the same bench.py command)"},
          open(os.path.join(sys.argv[1], "meta.json"), "w"))
PY
# the issue-rate microbenchmark (built here when the binary did not travel: hipcc is in the image)
[ -x $ROOT/tools/microbench/bin/valu_rates2 ] || { mkdir -p $ROOT/tools/microbench/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tools/microbench/valu_rates2.hip -o $ROOT/tools/microbench/bin/valu_rates2 2>/dev/null; }
$ROOT/tools/microbench/bin/valu_rates2 > $OUT/valu_rates2.txt 2>&1
cd /tmp && export TMPDIR=/tmp
# the profiled command: the bench's own loop after its spin-up (--spinup 50: the device at its sustained clock, as in the line), 20 timed frames
BENCH="python $ROOT/bench.py --steps 20 --warmup 2 --spinup 50 --no-cpu-baseline --no-stream-copy --no-cold --no-kernel-loops --no-configs4"
# 1. the bench line itself (un-profiled, with the CPU baselines)
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
# 2. kernel trace + stats
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $BENCH > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
# 3. counters, one set per pass (never combined with other trace domains), over the bench command itself.  (Round 3's profile took them over
#    tools/quick_time.py — same kernels and arguments, no torch — because bench.py "stopped answering" under --pmc: its scene pool's forked
#    workers inherited rocprofv3's SIGTERM handler and never finished Pool.terminate(); fixed in rfx_amd/scene.py, and the bench command's
#    FETCH_SIZE agrees with that collection to 0.1 % (K2: 2 %): profiles/r03_final/fetch_size_bench_command.csv.)
PMC_TARGET="$BENCH"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32" "SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p --output-format csv -- $PMC_TARGET > $OUT/pmc$i.log 2>&1 || echo "pmc set $i failed: $set" >> $OUT/errors.txt
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rocclr" in k or "Cijk" in k or "at::" in k or "elementwise" in k:
            continue
        acc[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
with open(out + "/pmc_hbm.csv", "w") as fh:
    w = csv.writer(fh); w.writerow(["counter", "kernel", "dispatches", "mean_value_KB"])
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in sorted(acc[c].items()):
            w.writerow([c, k, len(v), round(sum(v) / len(v), 1)])
with open(out + "/pmc_sq_l2.csv", "w") as fh:
    w = csv.writer(fh); w.writerow(["kernel", "counter", "dispatches", "mean_value"])
    for c in sorted(acc):
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        for k, v in sorted(acc[c].items()):
            w.writerow([k, c, len(v), "%.4g" % (sum(v) / len(v))])
PY
# the issue model, twice: the first pass yields every kernel's measured VALU instructions per pixel, which tools/isa_mix.py needs to weight the
# kernel's loop blocks; the second prices "int" / "other" with that kernel's own mix (isa_other_mix.json)
python $ROOT/tools/issue_model.py $OUT > $OUT/issue_model_flat_rates.txt 2>&1
python - "$OUT" "$ROOT" <<'PY'
import subprocess, sys
sys.path.insert(0, sys.argv[2] + "/tools")
import issue_model as IM
rows = IM.table(sys.argv[1])
args = []
for frag, key in IM.MIX_KEY:
    for name, m in rows.items():
        if frag in name:
            args.append("%s=%s" % (key, m["valu_per_px"]))
            break
subprocess.call([sys.executable, sys.argv[2] + "/tools/isa_mix.py", "--out", sys.argv[1] + "/isa_other_mix.json", "--valu-per-px"] + args,
                stdout=open(sys.argv[1] + "/isa_other_mix.txt", "w"), stderr=subprocess.STDOUT)
PY
python $ROOT/tools/issue_model.py $OUT > $OUT/issue_model.txt 2>&1
rm -rf $OUT/trace $OUT/pmc[0-9]* 
ls -la $OUT
tail -1 $OUT/bench.json | cut -c1-300
