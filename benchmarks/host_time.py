#!/usr/bin/env python
"""Where does the HOST time of a training step go?  Wall-clock of each enqueue phase (no device sync inside a step)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(dev)
opt = FlatAdam(m.parameters(), lr=1e-3)
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(dev), lens.to(dev), src.to(dev)
acc = {}


def tick(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def step(rec):
    t = time.perf_counter()
    opt.zero_grad(); t = tick("zero_grad", t) if rec else t
    est = m(mix); t = tick("forward", t) if rec else t
    loss = ctn.cal_loss(src, est, lens)[0]; t = tick("loss", t) if rec else t
    loss.backward(); t = tick("backward", t) if rec else t
    opt.step(max_grad_norm=5.0); t = tick("optim", t) if rec else t


for _ in range(3):
    step(False)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step(True)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host issue %.2f ms/step, wall %.2f ms/step" % (1e3 * t_issue / n, 1e3 * t_all / n))
for k, v in acc.items():
    print("  %-10s %.3f ms/step" % (k, 1e3 * v / n))
# the same with a device sync between steps: pure host cost of enqueueing into an EMPTY queue
acc.clear()
for _ in range(n):
    step(True)
    torch.cuda.synchronize()
print("with a sync after every step (queues never fill):")
for k, v in acc.items():
    print("  %-10s %.3f ms/step" % (k, 1e3 * v / n))
