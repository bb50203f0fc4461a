#!/usr/bin/env python
"""tests/golden/paper_traj_fp64.npz: the paper config trained for 10 optimiser steps on the bench's batch by the CPU oracle IN
FP64 (benchmarks/traj_fp64_oracle.py -> benchmarks/_traj64.pt, ~40 minutes of CPU) -- per-step losses, and every 997th element
of the initial and the final parameter vector (model.named_parameters() order), the yardstick of
tests/test_gpu_h3.py::test_trajectories_against_the_fp64_oracle.  usage: python oracle/make_traj_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import conv_tasnet_amd as ctn  # noqa: E402

STRIDE = 997
t = torch.load(os.path.join(ROOT, "benchmarks", "_traj64.pt"), weights_only=True)
torch.manual_seed(0)
m0 = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2)
p0 = torch.cat([p.detach().reshape(-1) for _, p in m0.named_parameters()])
assert p0.numel() == t["params"].numel()
np.savez(os.path.join(ROOT, "tests", "golden", "paper_traj_fp64.npz"), losses=np.asarray(t["losses"], dtype=np.float64),
         p0=p0[::STRIDE].numpy(), p_final=t["params"][::STRIDE].numpy(), stride=np.int64(STRIDE), steps=np.int64(t["steps"]), M=np.int64(t["M"]),
         travelled=np.float64((t["params"].double() - p0.double()).norm()))
print("losses", t["losses"])
