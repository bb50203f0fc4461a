#!/usr/bin/env python
"""Per-workgroup phase timeline of one split-bf16 GEMM form (needs a library built with -DCTN_EXP_B3_TIMELINE:
CTN_LIB_PATH=benchmarks/lab_gemm_TIMELINE.so).  Stamps: 0 entry, 1 first k-tile staged, 2 main loop done, 3 epilogue done;
s_memrealtime ticks are 10 ns, s_memtime counts shader cycles.  usage: gemm_timeline.py K1|K3|B1|B5 [warm launches]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

sys.argv = [sys.argv[0]] + sys.argv[1:]
form = sys.argv[1] if len(sys.argv) > 1 else "K1"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
saved = sys.argv
sys.argv = [saved[0], form, "0"]
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_only.py")).read().split("fn = fns[form]")[0]
ns = {"__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_only.py")}
exec(compile(src, "gemm_only_setup", "exec"), ns)
fn = ns["fns"][form]
for _ in range(warm):
    fn()
torch.cuda.synchronize()
fn()
torch.cuda.synchronize()
n = 8192 * 12
buf = (ctypes.c_ulonglong * n)()
rc = ctn.lib.load().ctn_debug_timeline(buf, n)
assert rc == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 12).astype(np.int64)
t = t[t[:, 1] > 0]
nwg = len(t)
rt = t[:, [1, 3, 5, 7]] * 10.0            # ns
mt = t[:, [0, 2, 4, 6]].astype(np.float64)
t0 = rt[:, 0].min()
rt -= t0
hw, xcc = t[:, 8], t[:, 9] & 0xf
cu = (xcc << 16) | (hw & 0xff00)          # cu_id[11:8], sh_id[12], se_id[15:13]
print("%s: %d workgroups on %d distinct (xcc, se, sh, cu); kernel span %.2f us" % (form, nwg, len(set(cu.tolist())), rt[:, 3].max() / 1e3))
clk = (mt[:, 3] - mt[:, 0]) / np.maximum(rt[:, 3] - rt[:, 0], 10.0)      # cycles per ns = GHz
print("in-kernel clock (cycles / ns over each workgroup's lifetime): median %.3f GHz, p10 %.3f, p90 %.3f" % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
names = ["fill (entry -> first k-tile staged)", "main loop", "epilogue"]
for i, nm in enumerate(names):
    d = (rt[:, i + 1] - rt[:, i]) / 1e3
    c = mt[:, i + 1] - mt[:, i]
    print("  %-38s median %6.2f us (p10 %5.2f p90 %5.2f)   %7.0f cycles" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90), np.median(c)))
start = rt[:, 0] / 1e3
print("  start times: first round (<1 us) %d workgroups; later %d; last start %.2f us" % ((start < 1.0).sum(), (start >= 1.0).sum(), start.max()))
# occupancy of each phase over time, 1-us bins
T = int(np.ceil(rt[:, 3].max() / 1e3)) + 1
print("  per-us count of workgroups in [fill, main, epilogue]:")
for us in range(T):
    a, b = us * 1e3, (us + 1) * 1e3
    row = []
    for i in range(3):
        ov = np.clip(np.minimum(rt[:, i + 1], b) - np.maximum(rt[:, i], a), 0, None).sum() / 1e3
        row.append(ov)
    print("    t=%2d us  fill %6.0f  main %6.0f  epi %6.0f" % (us, row[0], row[1], row[2]))
# co-residency: workgroups per CU in the first round
from collections import Counter
first = Counter(cu[start < 1.0].tolist())
print("  first-round workgroups per CU: ", sorted(Counter(first.values()).items()))
idx = np.argsort(t[:, 1])
order = [(int(cu[i] >> 16), int((hw[i] >> 13) & 7), int((hw[i] >> 12) & 1), int((hw[i] >> 8) & 15)) for i in idx[:24]]
print("  (xcc, se, sh, cu) of the first 24 workgroups by start time:", order)
bidx = np.nonzero(t[:, 1] > 0)[0]
