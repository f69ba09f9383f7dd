#!/usr/bin/env python
"""Largest idle gaps of every HIP queue inside the last full step of a rocprofv3 kernel trace, with the kernels on
either side and what the OTHER queues were running meanwhile (who is the step waiting for?)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_sgd")]
lo, hi = sg[-4] + 1, sg[-1] + 1
sel = rows[lo:hi]
t0 = int(sel[0]["Start_Timestamp"])
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    gaps = []
    for a, b in zip(ks, ks[1:]):
        if b[0] - a[1] > 20000:
            gaps.append((b[0] - a[1], a, b))
    gaps.sort(reverse=True)
    print("queue %s: %d kernels, %d gaps > 20 us, total %.2f ms" % (q, len(ks), len(gaps), sum(g[0] for g in gaps) / 1e6))
    for g, a, b in gaps[:8]:
        others = collections.Counter()
        for q2, k2 in byq.items():
            if q2 == q:
                continue
            for s, e, n in k2:
                ov = min(e, b[0]) - max(s, a[1])
                if ov > 0:
                    others[(q2, n)] += ov
        top = ", ".join("%s:%s %.0fus" % (k[0], k[1][-24:], v / 1e3) for k, v in others.most_common(3))
        print("   %7.1f us at %6.2f ms  after %-32s before %-32s | meanwhile: %s" % (g / 1e3, (a[1] - t0) / 1e6, a[2][-32:], b[2][-32:], top))
