#!/bin/bash
# round 2, GPU call A: whole -m gpu suite (incl. the BASELINE-config parity tests), XCD-group variants of K2/K3, HBM counters of the default build
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r02_a; mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 ) > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
bash tools/time_variants.sh > $OUT/variants.log 2>&1
grep -E "^==|K2 temporal|K3 pass|K1\+K2|sha1" $OUT/variants.log | head -80
# HBM-side traffic of the default build (one counter set per pass)
BENCH="python $PWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OLDPWD/$OUT/pmc_$set -o p --output-format csv -- $BENCH > $OLDPWD/$OUT/pmc_$set.log 2>&1
done
cd $OLDPWD
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r02_a/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]][r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for c in acc:
    for k, v in sorted(acc[c].items()):
        if "k1_" in k or "k2_" in k or "k3_" in k or "k4_" in k:
            print(c, k, len(v), round(sum(v) / len(v) / 1024, 1), "MiB")
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
