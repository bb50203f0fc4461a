#!/usr/bin/env python
"""Launch only the fused depthwise backward (B3) at the paper shape, for rocprofv3 --pmc / timing.
usage: python benchmarks/dw_bwd_only.py [dilation]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conv_tasnet_amd.ops import _p  # noqa: E402

dil = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M, H, K, P = 8, 512, 3199, 3
Kp = ops.padded_frames(K)
dev = "cuda:0"
f = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
dn2, d, h1 = f(M, H, Kp), f(M, H, Kp), f(M, H, Kp)
for t in (dn2, d, h1):
    t[..., K:] = 0
D = f(H, 1, P) * 0.3
g1, b1, g2 = f(1, H, 1), f(1, H, 1), f(1, H, 1)
a1 = torch.full((1,), 0.25, device=dev)
a2 = torch.full((1,), 0.2, device=dev)
ms1 = torch.tensor([[0.1, 1.2]] * M, device=dev)
ms2 = torch.tensor([[-0.05, 0.9]] * M, device=dev)
np2 = 400
s2p = torch.randn(M, np2, 2, device=dev, dtype=torch.float64)
Fr = ctn.lib.ctn_dw_bwd_rows(P, 1)
pc = torch.empty((Fr, M, H), device=dev)
s1p = torch.empty((M, H, 2), device=dev, dtype=torch.float64)
dn1 = torch.empty((M, H, Kp), device=dev)


def run():
    ctn.lib.call("ctn_dw_bwd", _p(dn2), _p(d), _p(h1), _p(dn1), _p(D), M, H, K, Kp, P, dil, 0, 1,
                 _p(g1), _p(b1), _p(a1), _p(ms1), _p(g2), _p(a2), _p(ms2), _p(s2p), np2, _p(pc), _p(s1p), ops._stream())


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("dw_bwd fused dil=%d: %.1f us  (%.2f TB/s on 4 x H*Kp*4*M bytes)" % (dil, us, 4 * M * H * Kp * 4 / us / 1e6))
