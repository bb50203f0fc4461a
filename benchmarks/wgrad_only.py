#!/usr/bin/env python
"""Launch only the weight-gradient GEMM (dW1 = dh1 . x^T, R=512, Cn=256) at the paper shape, for rocprofv3 --pmc."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

M, B, H, K = 8, 256, 512, 3199
Kp = ops.padded_frames(K)
xB = torch.randn(M, B, Kp, device="cuda:0"); xB[..., K:] = 0
xH = torch.randn(M, H, Kp, device="cuda:0"); xH[..., K:] = 0
for _ in range(12):
    ops.pw_wgrad(xH, xB, H, B, K)
torch.cuda.synchronize()
print("done")
