#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel name from a counter_collection CSV dir."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and pat not in k:
            continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s n=%3d avg=%.4g" % (c, len(v), sum(v) / len(v)))
