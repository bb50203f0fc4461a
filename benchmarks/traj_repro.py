#!/usr/bin/env python
"""Is a free-running trajectory reproducible?  The same 7 optimiser steps of the paper config (bench batch) under one arithmetic,
three times in one process -- the third time after the caching allocator has been dirtied with NaN-filled blocks of every size the
step uses -- printing the per-step losses with all digits and a checksum of the parameters.  usage: traj_repro.py [arith] [steps]"""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

DEV = "cuda:0"
arith = sys.argv[1] if len(sys.argv) > 1 else "h3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ctn.set_gemm_arith(arith)
mix, lens, src = O.synth_batch(0, 8, 32000)
mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)


def run():
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    out = []
    for _ in range(steps):
        opt.zero_grad()
        loss = ctn.cal_loss(src, m(mix), lens)[0]
        loss.backward()
        opt.step(max_grad_norm=5.0)
        out.append(float(loss.detach()))
    torch.cuda.synchronize()
    crc = zlib.crc32(opt.flat_params.detach().cpu().numpy().tobytes())
    del m, opt
    return out, crc


a = run()
b = run()
# dirty the allocator: free blocks of many sizes filled with NaN, so that any read of never-written memory shows
junk = [torch.full((n,), float("nan"), device=DEV) for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26, 13107200, 26214400, 6553600, 52428800)]
junk += [torch.full((8, 512, 3200), float("nan"), device=DEV) for _ in range(12)] + [torch.full((8, 256, 3200), float("nan"), device=DEV) for _ in range(12)]
del junk
c = run()
print(arith, "run 1:", " ".join("%.9f" % v for v in a[0]), "crc %08x" % a[1])
print(arith, "run 2:", " ".join("%.9f" % v for v in b[0]), "crc %08x" % b[1])
print(arith, "run 3 (allocator dirtied with NaN):", " ".join("%.9f" % v for v in c[0]), "crc %08x" % c[1])
print("reproducible:", a == b == c)
