#!/bin/bash
# round 3, call I2: bench.py under `rocprofv3 --pmc` after the pool-worker SIGTERM fix (rfx_amd/scene.py) — FETCH_SIZE / WRITE_SIZE per kernel of the
# bench command itself, next to the numbers collected over tools/quick_time.py in profiles/r03_final/pmc_hbm.csv
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r03_i2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stream-copy"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- $BENCH > $OUT/$c.log 2>&1; echo "$c rc=$?" | tee -a $OUT/rc.txt
done
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rocclr" in k or "Cijk" in k or "at::" in k or "elementwise" in k: continue
        acc[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
with open(out + "/pmc_hbm_bench.csv", "w") as fh:
    w = csv.writer(fh); w.writerow(["counter", "kernel", "dispatches", "mean_value_KB"])
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in sorted(acc[c].items()):
            w.writerow([c, k, len(v), round(sum(v) / len(v), 1)])
print(open(out + "/pmc_hbm_bench.csv").read()[:3000])
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
tail -3 $OUT/FETCH_SIZE.log | cut -c1-300
