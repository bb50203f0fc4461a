#!/usr/bin/env python
"""Sweep the weight-gradient GEMM's tile / split plan at the paper shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from gemm_sweep import timeit, M, K, Kp  # noqa: E402

dev = "cuda:0"
B, H = 256, 512
xB = torch.randn(M, B, Kp, device=dev); xB[..., K:] = 0
xH = torch.randn(M, H, Kp, device=dev); xH[..., K:] = 0
a = torch.full((1,), 0.25, device=dev)
g = torch.randn(1, H, 1, device=dev)
b = torch.randn(1, H, 1, device=dev)
ms = torch.tensor([[0.1, 1.3]] * M, device=dev)
flop = 2.0 * H * B * M * K
ref = torch.einsum("mrk,mck->rc", xH[:2].double().cpu(), xB[:2].double().cpu())
for tile in (64, 12864, 128):
    for blocks in (256, 384, 512, 768, 1024):
        ctn.lib.ctn_tune_wgrad(tile, blocks)
        got = ops.pw_wgrad(xH[:2].contiguous(), xB[:2].contiguous(), H, B, K)
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        t1 = timeit(lambda: ops.pw_wgrad(xH, xB, H, B, K))
        t2 = timeit(lambda: ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms)))
        ws = ctn.lib.ctn_pw_wgrad_workspace(M, H, B, Kp) / 2**20
        print("tile %5d blocks %4d: plain %6.1f us (%5.1f TF)  pro %6.1f us  slabs %5.1f MiB  err %.1e" % (tile, blocks, t1, flop / t1 / 1e6, t2, ws, err), flush=True)
