#!/usr/bin/env python
"""One training step's gradients of the paper config under h3 / b6 / fp32 MFMA against the fp64 oracle gradient
(benchmarks/_grad64_m<M>.pt from oracle/make_grad_golden.py), split by parameter family: where does each arithmetic's error
sit, and what makes the total grow with the batch?  usage: python benchmarks/arith_grad_families.py M [M ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

DEV = "cuda:0"


def family(k):
    if k.startswith("encoder") or k.startswith("decoder"):
        return k.split(".")[0]
    p = k.split(".")
    if k.startswith("separator.network.0"):
        return "input cLN"
    if k.startswith("separator.network.1"):
        return "bottleneck 1x1"
    if k.startswith("separator.network.3"):
        return "mask 1x1"
    tail = ".".join(p[5:])
    return {"net.0.weight": "block 1x1 (B->H)", "net.1.weight": "PReLU 1", "net.2.gamma": "gLN 1 gamma", "net.2.beta": "gLN 1 beta",
            "net.3.net.0.weight": "depthwise", "net.3.net.1.weight": "PReLU 2", "net.3.net.2.gamma": "gLN 2 gamma",
            "net.3.net.2.beta": "gLN 2 beta", "net.3.net.3.weight": "block 1x1 (H->B)"}.get(tail, tail)


for M in [int(a) for a in sys.argv[1:]] or [8]:
    ref = torch.load(os.path.join(ROOT, "benchmarks", "_grad64_m%d.pt" % M), weights_only=True)
    g64 = ref["grad"].double()
    mix, lens, src = O.synth_batch(0, M, 32000)
    print("== M = %d: fp64 loss %.9f |g64| %.4e" % (M, ref["loss"], float(g64.norm())))
    for arith in ("h3", "b6", "fp32"):
        ctn.set_gemm_arith(arith)
        torch.manual_seed(0)
        m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
        loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
        loss.backward()
        torch.cuda.synchronize()
        off, fam, tot = 0, {}, 0.0
        for k, p in m.named_parameters():
            n = p.numel()
            d = p.grad.detach().double().cpu().reshape(-1) - g64[off:off + n]
            e, r = float((d ** 2).sum()), float((g64[off:off + n] ** 2).sum())
            a, b = fam.get(family(k), (0.0, 0.0))
            fam[family(k)] = (a + e, b + r)
            tot += e
            off += n
        print("   %-4s loss err %.1e dB; |g - g64| / |g64| = %.3e; by family (share of the squared error, relative error of the family): %s" %
              (arith, abs(float(loss.detach()) - ref["loss"]), tot ** 0.5 / float(g64.norm()),
               "; ".join("%s %.0f%% (%.1e)" % (f, 100 * e / tot, (e / max(r, 1e-300)) ** 0.5) for f, (e, r) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:6])))
ctn.set_gemm_arith("h3")
