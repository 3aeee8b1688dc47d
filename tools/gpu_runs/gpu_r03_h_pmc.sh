#!/bin/bash
# round 3, call H2: SQ counters of the shipped K1 and of the compacting variant c5 (csrc/variants, built from profiles/r03_experiments/k1_compacting_march.patch)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r03_h_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in old c5; do
  if [ $v = old ]; then LIB=$ROOT/realism-effects_amd/csrc/librfx_hip.so; else LIB=$ROOT/realism-effects_amd/csrc/variants/librfx_k1_ssgi_$v.so; fi
  i=0
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/$v$i -o p --output-format csv -- python $ROOT/tools/quick_time.py --lib $LIB 3840 2160 3 K1 > $OUT/$v$i.log 2>&1 || echo "pass $v $i failed" >> $OUT/errors.txt
  done
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
with open(out + "/summary.txt", "w") as fh:
    for v in ("old", "c5"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(out + "/%s[0-9]/**/*counter_collection.csv" % v, recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "k1_ssgi_march" not in k: continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in sorted(acc.items()):
            line = "%s %s: " % (v, k[:60]) + "  ".join("%s=%.4g(n%d)" % (c, sum(x) / len(x), len(x)) for c, x in sorted(cs.items()))
            print(line); fh.write(line + "\n")
PY
rm -rf $OUT/old[0-9] $OUT/c5[0-9]
cat $OUT/errors.txt 2>/dev/null
