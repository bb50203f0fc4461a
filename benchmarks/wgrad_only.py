#!/usr/bin/env python
"""Launch only one weight-gradient GEMM at the paper shape, for rocprofv3 --pmc: dW1 = dh1 . x^T (R=512, Cn=256; default) or,
with the argument `pro`, dW2 = dout . gLN2(prelu(d))^T (R=256, Cn=512, fused prologue)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

M, B, H, K = 8, 256, 512, 3199
Kp = ops.padded_frames(K)
xB = torch.randn(M, B, Kp, device="cuda:0"); xB[..., K:] = 0
xH = torch.randn(M, H, Kp, device="cuda:0"); xH[..., K:] = 0
pro = len(sys.argv) > 1 and sys.argv[1] == "pro"
g = torch.randn(1, H, 1, device="cuda:0")
b = torch.randn(1, H, 1, device="cuda:0")
a = torch.full((1,), 0.25, device="cuda:0")
ms = torch.tensor([[0.1, 1.3]] * M, device="cuda:0")
for _ in range(12):
    if pro:
        ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms))
    else:
        ops.pw_wgrad(xH, xB, H, B, K)
torch.cuda.synchronize()
print("done")
