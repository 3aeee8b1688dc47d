#!/bin/bash
# round 3, call I3: bench.py under `rocprofv3 --pmc FETCH_SIZE` once the scene pool's workers leave without SIGTERM (faulthandler after 50 s if it still stops)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/r03_i3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WRAP="import faulthandler, sys, runpy; faulthandler.dump_traceback_later(50, exit=True); sys.argv = ['bench.py', '--steps', '5', '--warmup', '1', '--no-cpu-baseline', '--no-stream-copy']; runpy.run_path('$ROOT/bench.py', run_name='__main__')"
timeout 80 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o p --output-format csv -- python -c "$WRAP" > $OUT/f.log 2>&1; echo "rc=$?" | tee $OUT/rc.txt
python - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rocclr" in k or "Cijk" in k or "at::" in k or "elementwise" in k: continue
        acc[k].append(float(r["Counter_Value"]))
with open(out + "/fetch_size_bench.csv", "w") as fh:
    w = csv.writer(fh); w.writerow(["counter", "kernel", "dispatches", "mean_value_KB"])
    for k, v in sorted(acc.items()): w.writerow(["FETCH_SIZE", k, len(v), round(sum(v) / len(v), 1)])
print(open(out + "/fetch_size_bench.csv").read()[:2500])
PY
rm -rf $OUT/f
grep -v "^W2026\|^E2026" $OUT/f.log | tail -25 | cut -c1-250
