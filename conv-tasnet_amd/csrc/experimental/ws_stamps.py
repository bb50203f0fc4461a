#!/usr/bin/env python
"""Diagnostic: phase shares of the wave-specialised split-bf16 GEMM (needs a -DCTN_WS_STAMPS build:
CTN_EXTRA_HIPCC_FLAGS=-DCTN_WS_STAMPS python -c 'import __graft_entry__ as g; g.build()')."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CTN_GEMM_MODE"] = "x6"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

M, B, H, K = 8, 256, 512, 3199
Kp = ops.padded_frames(K)
ctn.lib.ctn_tune_pw_tile(10)
x = torch.randn(M, B, Kp, device="cuda:0")
x[..., K:] = 0
W = torch.randn(H, B, device="cuda:0") * 0.05
for _ in range(5):
    ops.pw_gemm(W, x, H, B, K)
torch.cuda.synchronize()
dll = ctypes.CDLL(ctn.LIB_PATH)
buf = (ctypes.c_ulonglong * 8192)()
assert dll.ctn_ws_debug_read(buf, 8192) == 0
d = np.array(buf, dtype=np.float64).reshape(512, 16)
names_c = ["loop", "compute", "barrier", "prologue_wait", "t_start", "epilogue", "t_end"]
print("s_memtime ticks (100 MHz constant clock: 1 tick = 10 ns); mean over the first 512 workgroups")
for i, n in enumerate(names_c):
    if n.startswith("t_"):
        continue
    print("consumer %-14s %10.1f" % (n, d[:, i].mean()))
for i, n in enumerate(["loop(steady)", "write_lds", "load_issue", "barrier"]):
    print("producer %-14s %10.1f" % (n, d[:, 8 + i].mean()))
print("workgroup lifetime (consumer wave)  %10.1f" % (d[:, 6] - d[:, 4]).mean())
