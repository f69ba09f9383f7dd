#!/usr/bin/env python
"""Steady-state per-kernel summary of a rocprofv3 kernel trace of bench.py (last 6 steps)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_sgd")]
per, nst = 3, 6
steps = len(ends) // per
lo = ends[(steps - nst) * per - 1] + 1
sel = rows[lo:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r["Kernel_Name"].split("(")[0][-64:]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print("# steady state, %d steps: %.2f ms of kernel time per step" % (nst, tot / nst / 1e3))
print("%-66s %8s %9s %8s" % ("kernel", "n/step", "ms/step", "avg_us"))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-66s %8.1f %9.3f %8.1f" % (n, v[0] / nst, v[1] / nst / 1e3, v[1] / v[0]))
