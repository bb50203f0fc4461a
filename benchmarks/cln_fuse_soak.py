#!/usr/bin/env python
"""Soak of the causal cLN config: N optimiser steps (FlatAdam, clip 5) on one fixed batch from the same initial weights under
ctn_tune("cln_fuse", 0) and (2): the losses every 25 steps, finite gradients throughout, and the two runs side by side (they separate
by rounding only: same mathematics).  usage: python benchmarks/cln_fuse_soak.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(dev), lens.to(dev), src.to(dev)
curves = {}
for level in (0, 2):
    ctn.lib.call("ctn_tune", b"cln_fuse", level)
    ctn.ops._ws_cache.clear()
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2, norm_type="cLN", causal=True).to(dev)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    out = []
    for it in range(N):
        opt.zero_grad()
        loss = ctn.cal_loss(src, m(mix), lens)[0]
        loss.backward()
        gn = opt.step(max_grad_norm=5.0)
        if it % 25 == 0 or it == N - 1:
            v = float(loss.detach())
            assert v == v and abs(v) < 1e6, (level, it, v)
            out.append((it, v))
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(p).all()) for p in m.parameters())
    curves[level] = out
ctn.lib.call("ctn_tune", b"cln_fuse", 2)
print("step   loss cln_fuse=0   loss cln_fuse=2   (dB, negative SI-SNR; fixed batch of 8 x 4 s)")
for (it, a), (_, b) in zip(curves[0], curves[2]):
    print("%4d   %12.5f   %12.5f" % (it, a, b))
