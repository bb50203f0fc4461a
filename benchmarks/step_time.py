#!/usr/bin/env python
"""Training-step time of a paper-size model variant.
usage: python benchmarks/step_time.py [gLN|cLN|BN] [causal 0|1] [C] [L] [samples]   (C3 config of BASELINE: gLN 0 3 16 64000)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402

norm = sys.argv[1] if len(sys.argv) > 1 else "gLN"
causal = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
C = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = int(sys.argv[4]) if len(sys.argv) > 4 else 20
T = int(sys.argv[5]) if len(sys.argv) > 5 else 32000
dev = "cuda:0"
torch.manual_seed(0)
m = ctn.ConvTasNet(256, L, 256, 512, 3, 8, 4, C, norm_type=norm, causal=causal).to(dev)
opt = FlatAdam(m.parameters(), lr=1e-3)
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=T, C=C, sample_rate=8000 * T // 32000)))
mix, lens, src = mix.to(dev), lens.to(dev), src.to(dev)


def step():
    opt.zero_grad()
    loss = ctn.cal_loss(src, m(mix), lens)[0]
    loss.backward()
    opt.step(max_grad_norm=5.0)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    step()
t_issue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("%s causal=%d C=%d L=%d T=%d: %.2f ms/step, %.1f utt/s (host issue %.2f ms/step)" % (norm, causal, C, L, T, dt * 1e3, 8 / dt, t_issue * 1e3))
