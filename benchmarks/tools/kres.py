#!/usr/bin/env python
"""Per-kernel resource usage + static instruction mix from a hipcc -S listing.  usage: kres.py file.s [name-substring]"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
meta = {}
for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S):
    meta[m.group(2)] = (int(m.group(1)), int(m.group(3)), int(m.group(4)))
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    lines = body.split("\n")
    valu = sum(1 for l in lines if re.match(r"\s+v_(?!mfma)", l))
    mfma = sum(1 for l in lines if "v_mfma" in l)
    lds = sum(1 for l in lines if re.match(r"\s+ds_", l))
    vm = sum(1 for l in lines if re.match(r"\s+(buffer_|global_|flat_)", l))
    lds_sz, sg, vg = meta.get(name, (0, 0, 0))
    print("%-90s lds %6d sgpr %3d vgpr %3d | static valu %4d mfma %3d lds %3d vmem %3d" % (name[-90:], lds_sz, sg, vg, valu, mfma, lds, vm))
