#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel avg us and ms per bench step."""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
f = (glob.glob(d + "/*/*_kernel_stats.csv") + glob.glob(d + "/*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.2f ms" % (tot / 1e6 / steps))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-78s calls %5s avg_us %8.1f ms/step %6.2f  %4.1f%%" % (n[:78], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Percentage"])))
