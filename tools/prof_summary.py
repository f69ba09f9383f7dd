#!/usr/bin/env python
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a small text summary
(committed under profiles/; the raw traces stay in gpurun_out/)."""
import csv
import glob
import sys


def main(d, steps, out):
    f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % f.split("/")[-1],
             "# %d profiled steps (warm-up included); total kernel time %.2f ms = %.2f ms/step"
             % (steps, tot / 1e6, tot / 1e6 / steps),
             "%-72s %7s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct")]
    for r in rows[:45]:
        lines.append("%-72s %7s %10.2f %10.1f %6.1f" % (r["Name"][:72], r["Calls"],
                                                         float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3,
                                                         float(r["Percentage"])))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:32]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
