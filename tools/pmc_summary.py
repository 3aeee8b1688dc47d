#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import csv, collections, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rocclr" in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print("%-60s " % k[:60] + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())) + "  (n=%d)" % len(next(iter(cs.values()))))
