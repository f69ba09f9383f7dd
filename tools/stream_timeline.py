#!/usr/bin/env python
"""Per-stream occupancy of one eager training step from a rocprofv3 kernel trace:
span, busy time and first / last kernel of every HIP stream (queue) inside the last full step."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_sgd")]
# a step ends with 3 k_sgd launches; take the span between the ends of the last two steps
lo, hi = sg[-4] + 1, sg[-1] + 1
sel = rows[lo:hi]
t0 = int(sel[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in sel)
print("step span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(sel)))
byq = collections.defaultdict(list)
for r in sel:
    byq[(r["Queue_Id"], r.get("Stream_Id", ""))].append(r)
allbusy = []
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    a, b = int(rs[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rs)
    print("queue %s: %4d kernels, busy %.2f ms, active from %.2f to %.2f ms  first=%s last=%s" % (
        q, len(rs), busy / 1e6, (a - t0) / 1e6, (b - t0) / 1e6, rs[0]["Kernel_Name"][:28], rs[-1]["Kernel_Name"][:28]))
    allbusy += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rs]
allbusy.sort()
cur_s, cur_e, union = allbusy[0][0], allbusy[0][1], 0
for s, e in allbusy[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("GPU busy (union of all kernels) %.2f ms = %.1f %% of the step" % (union / 1e6, 100.0 * union / (t1 - t0)))

# idle time (no kernel of any stream running), attributed to the kernel that ENDS the gap
gaps = collections.defaultdict(lambda: [0, 0.0])
evs = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in sel])
cur_e = evs[0][1]
half = t0 + (t1 - t0) // 2
idle_half = [0.0, 0.0]
for s, e, name in evs[1:]:
    if s > cur_e:
        g = gaps[name]
        g[0] += 1
        g[1] += (s - cur_e) / 1e3
        idle_half[0 if s < half else 1] += (s - cur_e) / 1e3
    cur_e = max(cur_e, e)
tot = sum(v[1] for v in gaps.values())
print("idle %.2f ms in %d gaps (first half of the step %.2f ms, second half %.2f ms); by the kernel that ends the gap:" % (
    tot / 1e3, sum(v[0] for v in gaps.values()), idle_half[0] / 1e3, idle_half[1] / 1e3))
for name, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("   %-46s %4d gaps %8.1f us  (%.1f us each)" % (name, v[0], v[1], v[1] / v[0]))
