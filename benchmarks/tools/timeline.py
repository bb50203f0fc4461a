#!/usr/bin/env python
"""Print the kernel timeline of one training step from a rocprofv3 --kernel-trace CSV (two queues side by side).
usage: timeline.py p_kernel_trace.csv [step_index] [first_row] [nrows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at each clip_adam_kernel end
ends = [i for i, r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 2
seg = rows[ends[k] + 1: ends[k + 1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(sys.argv[4]) if len(sys.argv) > 4 else 60
queues = sorted(set(r["Queue_Id"] for r in seg))


def short(nm):
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"(\w+)(<[^>]*>)?", nm)
    return (m.group(1) + (m.group(2) or ""))[:34]


print("step %d: %d kernels, %.2f ms, queues %s" % (k, len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e6, queues))
for r in seg[first:first + n]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    col = queues.index(r["Queue_Id"])
    print("%9.1f %9.1f %7.1f  %s%s" % (s, e, e - s, " " * (38 * col), short(r["Kernel_Name"])))
