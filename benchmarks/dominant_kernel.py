#!/usr/bin/env python
"""Launch only the dominant kernel (K1: 1x1 conv B->H, fp32 MFMA, PReLU-stats epilogue) at the bench shape.
Used under rocprofv3 --pmc to read MFMA-busy and HBM-traffic counters for that kernel alone."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

M, B, H, K = 8, 256, 512, 3199
Kp = ops.padded_frames(K)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else -1
ctn.lib.ctn_tune_pw_tile(tile)
x = torch.randn(M, B, Kp, device="cuda:0")
x[..., K:] = 0
W = torch.randn(H, B, device="cuda:0") * 0.05
a = torch.full((1,), 0.25, device="cuda:0")
for _ in range(12):
    ops.pw_gemm(W, x, H, B, K, epi_alpha=a)
torch.cuda.synchronize()
print("done")
