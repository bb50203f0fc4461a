#!/usr/bin/env python
"""Extract one kernel from a hipcc -S --cuda-device-only listing and count its instructions per basic block.
usage: python benchmarks/tools/asm_fn.py file.s <mangled-name-substring> [--dump]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
dump = "--dump" in sys.argv
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.endswith(":") is False and re.match(r"^_Z\S*:", l) and pat in l:
        start = i
        break
if start is None:
    raise SystemExit("kernel not found")
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
print(lines[start].split(":")[0][:160])


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return op.split()[0]
    if op.startswith("s_"):
        return "salu"
    return "other"


blocks, cur, name = [], collections.Counter(), "entry"
ops = collections.Counter()
for l in body[1:]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append((name, cur))
            cur, name = collections.Counter(), t.split(":")[0]
        continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append((name, cur))
        cur, name = collections.Counter(), t.split(":")[0]
        continue
    op = t.split()[0]
    cur[cls(op)] += 1
    if cls(op) == "valu":
        ops[op] += 1
    if dump:
        print(l)
blocks.append((name, cur))
for n, c in blocks:
    tot = sum(c.values())
    if tot >= 20:
        print("%-12s total %4d  " % (n, tot) + "  ".join("%s %d" % kv for kv in sorted(c.items())))
print("VALU opcodes:", ", ".join("%s %d" % kv for kv in ops.most_common(25)))
