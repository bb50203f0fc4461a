#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr saved to a
file).  usage: python benchmarks/tools/kres.py res.txt [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
blocks = txt.split("remark: Function Name: ")[1:]
names = [b.split(" [")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, n in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(int\)", "", n)
    if pats and not all(p in n for p in pats):
        continue
    print("%-110s vgpr %3d agpr %3d spill %d occ %d lds %6d sgpr %3d" % (n[:110], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"),
          g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), g("SGPRs")))
