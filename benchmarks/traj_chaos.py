#!/usr/bin/env python
"""How fast do rounding-level differences grow over a training run?  10 optimiser steps of the paper config (bench batch) under
each arithmetic, and under the SAME arithmetic with a different (equally valid) summation order of the weight gradients (the
split-K plan: ctn_tune wgrad_blocks / b3_wgrad_blocks).  Everything is compared with the default fp32-MFMA run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import conv_tasnet_amd as ctn
from conv_tasnet_amd import ops
from conv_tasnet_amd.optim import FlatAdam
from conv_tasnet_amd.train import SyntheticLoader
DEV = "cuda:0"
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
def run(arith, key=None, val=None, steps=10):
    ctn.set_gemm_arith(arith)
    if key:
        ctn.lib.call("ctn_tune", key, val)
        ops._ws_cache.clear()
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    p0 = opt.flat_params.detach().clone()
    ls = []
    for _ in range(steps):
        opt.zero_grad()
        loss = ctn.cal_loss(src, m(mix), lens)[0]
        loss.backward()
        opt.step(max_grad_norm=5.0)
        ls.append(float(loss.detach()))
    return ls, opt.flat_params.detach().clone(), p0
l0, p0f, pinit = run("fp32")
trav = float((p0f - pinit).double().norm())
print("fp32 default: losses %s; distance travelled %.4e" % (" ".join("%.6f" % v for v in l0), trav))
for arith, key, val, restore in (("fp32", b"wgrad_blocks", 300, 512), ("fp32", b"wgrad_blocks", 1024, 512), ("b6", None, None, None), ("b6", b"b3_wgrad_blocks", 200, 256),
                                 ("h3", None, None, None), ("h3", b"b3_wgrad_blocks", 200, 256), ("h3", b"b3_wgrad_blocks", 512, 256)):
    l, p, _ = run(arith, key, val)
    if key:
        ctn.lib.call("ctn_tune", key, restore)
        ops._ws_cache.clear()
    print("%-5s %-22s max |loss - fp32 default| %.2e dB   |p - p_fp32| / travelled %.3e" % (arith, "" if not key else "%s=%d" % (key.decode(), val),
          max(abs(a - b) for a, b in zip(l, l0)), float((p - p0f).double().norm()) / trav), flush=True)
ctn.set_gemm_arith("h3")
