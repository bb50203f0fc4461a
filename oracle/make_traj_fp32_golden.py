#!/usr/bin/env python
"""tests/golden/paper_traj_cpu_fp32.npz: the paper config trained for 10 optimiser steps on the bench's batch (M = 8 x 4 s) by the
CPU oracle in FP32 -- stock PyTorch CPU ops, the reference's own arithmetic (oracle/ctn_oracle.py is pinned to the imported
reference by tests/golden/model_*.npz and solver_traj*.npz) -- per-step losses and every 997th element of the initial and final
parameter vector.  Yardstick of tests/test_gpu_h3.py::test_trajectories_against_the_reference_arithmetic: does a HIP arithmetic
follow the REFERENCE arithmetic's trajectory?  (paper_traj_fp64.npz asks how it follows exact arithmetic.)
usage: python oracle/make_traj_fp32_golden.py        (~8 minutes on 8 cores; the thread count changes summation orders, hence
last bits: the test's bounds allow for that)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import conv_tasnet_amd as ctn  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

STRIDE, STEPS, M = 997, 10, 8
cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
torch.manual_seed(0)
m0 = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C)
names = [k for k, _ in m0.named_parameters()]
sd = {k: v.detach().clone() for k, v in m0.state_dict().items()}
p0 = torch.cat([sd[k].reshape(-1) for k in names]).clone()
mix, lens, src = O.synth_batch(0, M, 32000)
st, losses = {}, []
for i in range(STEPS):
    t0 = time.time()
    losses.append(O.train_step(cfg, sd, st, mix, src, lens))
    print("cpu fp32 step %d loss %.6f (%.0f s)" % (i, losses[-1], time.time() - t0), flush=True)
p1 = torch.cat([sd[k].reshape(-1) for k in names])
np.savez(os.path.join(ROOT, "tests", "golden", "paper_traj_cpu_fp32.npz"), losses=np.asarray(losses, dtype=np.float64),
         p0=p0[::STRIDE].numpy(), p_final=p1[::STRIDE].numpy(), stride=np.int64(STRIDE), steps=np.int64(STEPS), M=np.int64(M),
         travelled=np.float64((p1.double() - p0.double()).norm()), threads=np.int64(torch.get_num_threads()))
