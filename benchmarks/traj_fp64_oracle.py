#!/usr/bin/env python
"""The CPU oracle in fp64: `steps` optimiser steps of the paper config on the bench's batch from the seeded initial weights.
Writes losses and the final parameters (name order of the model) to benchmarks/_traj64.pt -- the yardstick for
benchmarks/traj_vs_fp64.py.  CPU only (minutes).  usage: python benchmarks/traj_fp64_oracle.py [steps] [M]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402   (the model class only gives the seeded initial weights; nothing runs on it here)
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
mix, lens, src = next(iter(SyntheticLoader(1, M, samples=32000)))
torch.manual_seed(0)
m0 = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C)
names = [k for k, _ in m0.named_parameters()]
sd = {k: v.detach().double().clone() for k, v in m0.state_dict().items()}
st, losses = {}, []
for i in range(steps):
    t0 = time.time()
    losses.append(O.train_step(cfg, sd, st, mix.double(), src.double(), lens))
    print("fp64 oracle step %d loss %.9f (%.0f s)" % (i, losses[-1], time.time() - t0), flush=True)
torch.save({"losses": losses, "params": torch.cat([sd[k].reshape(-1) for k in names]).float(), "steps": steps, "M": M},
           os.path.join(ROOT, "benchmarks", "_traj64.pt"))
