#!/usr/bin/env python
"""Mean PMC counter values per kernel from a rocprofv3 --pmc csv directory."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[name]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, cs in agg.items():
            if len(cs["_dur_us"]) < 10:
                continue
            print(k)
            for c, v in sorted(cs.items()):
                print("    %-28s %.4g" % (c, sum(v[len(v)//4:]) / max(1, len(v[len(v)//4:]))))
