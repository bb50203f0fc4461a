#!/usr/bin/env python
"""Launch only the fused depthwise forward (K2: gLN1 + PReLU prologue, depthwise, PReLU2 / gLN2 statistics) at the paper shape,
whole batch and half batch, for timing / rocprofv3 --pmc.  usage: python benchmarks/dw_fwd_only.py [dilation] [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

dil = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Ms = [int(sys.argv[2])] if len(sys.argv) > 2 else [8, 4]
H, K, P = 512, 3199, 3
Kp = ops.padded_frames(K)
dev = "cuda:0"
for M in Ms:
    h1 = torch.randn(M, H, Kp, device=dev); h1[..., K:] = 0
    D = torch.randn(H, 1, P, device=dev) * 0.3
    g1, b1 = torch.randn(1, H, 1, device=dev), torch.randn(1, H, 1, device=dev)
    a1 = torch.full((1,), 0.25, device=dev)
    a2 = torch.full((1,), 0.2, device=dev)
    part = torch.randn(M, 200, 2, device=dev, dtype=torch.float64).abs() * 1e3      # K1's statistics partials (4 row tiles x 50 column tiles)
    ms = torch.empty(M, 2, device=dev)

    def run():
        return ops.dw_fwd(h1, D, K, dil, False, pro=(part, g1, b1, a1), epi_alpha=a2, ms_out=ms)

    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("dw_fwd fused dil=%d M=%d: %.1f us  (%.2f TB/s on 2 x M*H*Kp*4 bytes)" % (dil, M, us, 2 * M * H * Kp * 4 / us / 1e6))
