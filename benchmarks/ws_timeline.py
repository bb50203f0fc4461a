#!/usr/bin/env python
"""Where the two roles of pw_gemm_ws_kernel spend their cycles (needs a library built with -DCTN_EXP_B3_TIMELINE:
CTN_LIB_PATH=benchmarks/lab_gemm_TIMELINE.so).  Per workgroup: cycles of the MFMA role (wave 0) in total / waiting at barriers /
writing patches, cycles of the IO role (wave 4) in total / waiting at barriers.  usage: ws_timeline.py K1|K3|B1|B5 [blocks]"""
import ctypes
import os
import sys

import numpy as np

form = sys.argv[1] if len(sys.argv) > 1 else "K1"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 512
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
sys.argv = [sys.argv[0], form, "0"]
import torch  # noqa: E402
src = open(os.path.join(here, "gemm_only.py")).read().split("fn = fns[form]")[0]
ns = {"__file__": os.path.join(here, "gemm_only.py")}
exec(compile(src, "gemm_only_setup", "exec"), ns)
ctn = ns["ctn"]
ctn.lib.call("ctn_tune", b"b3_ws_blocks", blocks)
fn = ns["fns"][form]
for _ in range(300):
    fn()
torch.cuda.synchronize()
fn()
torch.cuda.synchronize()
n = 8192 * 12
buf = (ctypes.c_ulonglong * n)()
assert ctn.lib.load().ctn_debug_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 12).astype(np.int64)[:blocks]
span = (t[:, 6] - t[:, 5]) * 10.0 / 1e3      # us (s_memrealtime: 100 MHz)
clk = t[:, 3] / np.maximum((t[:, 6] - t[:, 5]) * 10.0, 10.0)
print("%s, %d workgroups: kernel span (first start .. last end) %.2f us; per-workgroup lifetime median %.2f us (p10 %.2f p90 %.2f); in-kernel clock %.3f GHz"
      % (form, blocks, (t[:, 6].max() - t[:, 5].min()) * 10.0 / 1e3, np.median(span), np.percentile(span, 10), np.percentile(span, 90), np.median(clk)))
for tiles in sorted(set(t[:, 7].tolist())):
    s = t[t[:, 7] == tiles]
    if len(s) == 0 or tiles == 0:
        continue
    med = lambda c: float(np.median(s[:, c]))
    print("  %d workgroups with %d tiles: table %6.0f cycles | MFMA role total %7.0f, at barriers %7.0f (%.0f %%), patch writes %6.0f | IO role total %7.0f, at barriers %7.0f (%.0f %%) | per tile %.0f cycles = %.0f per k-tile"
          % (len(s), tiles, med(8), med(0), med(1), 100 * med(1) / med(0), med(2), med(3), med(4), 100 * med(4) / med(3), med(0) / tiles, med(0) / tiles / (8 if form in ("K1", "B1") else 16)))
