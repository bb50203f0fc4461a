#!/usr/bin/env python
"""Causal cLN config (paper widths): one step's loss and gradients under ctn_tune("cln_fuse", 0 | 1 | 2) and every arithmetic against
the CPU oracle run in fp64 on the same weights and batch -- does taking the channel-wise norms' statistics / backward sums out of
the GEMM epilogues (column sums, single-pass variance in fp64) cost accuracy against the stand-alone two-pass kernels?
usage: python benchmarks/cln_fuse_grad_err.py [M] [T]      (default 2 utterances of 1 s: the fp64 oracle step takes ~1 min)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2, norm_type="cLN", causal=True)
torch.manual_seed(0)
m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2, norm_type="cLN", causal=True).to(DEV)
mix, lens, src = O.synth_batch(0, M, T)
t0 = time.time()
sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in m.state_dict().items()}
loss64 = O.cal_loss(src.double(), O.forward(cfg, sd, mix.double()), lens)[0]
loss64.backward()
names = [k for k, _ in m.named_parameters()]
g64 = torch.cat([sd[k].grad.reshape(-1) for k in names])
print("fp64 oracle: loss %.9f |g64| %.4e (%.0f s)" % (float(loss64), float(g64.norm()), time.time() - t0), flush=True)
for arith in ("h3", "b6", "fp32"):
    ctn.set_gemm_arith(arith)
    for level in (0, 1, 2):
        ctn.lib.call("ctn_tune", b"cln_fuse", level)
        m.zero_grad()
        loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
        loss.backward()
        ctn.ops.join_side_stream()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.detach().double().cpu().reshape(-1) for p in m.parameters()])
        worst, wk, off = 0.0, "", 0
        gmax = float(g64.abs().max())
        for k, p in m.named_parameters():
            n = p.numel()
            e = float((g[off:off + n] - g64[off:off + n]).abs().max()) / max(float(g64[off:off + n].abs().max()), 1e-3 * gmax)
            if e > worst:
                worst, wk = e, k
            off += n
        print("  %-4s cln_fuse=%d  loss err %.2e dB  |g - g64| / |g64| = %.3e  worst tensor (max-norm) %.2e %s" %
              (arith, level, abs(float(loss.detach()) - float(loss64)), float((g - g64).norm() / g64.norm()), worst, wk), flush=True)
ctn.lib.call("ctn_tune", b"cln_fuse", 2)
ctn.set_gemm_arith("h3")
