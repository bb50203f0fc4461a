#!/usr/bin/env python
"""tests/golden/paper_grad_fp64_m8.npz: ONE training step's gradients of the paper config on the bench's batch (M = 8 x 4 s) by the
CPU oracle IN FP64 (the build's restatement of the reference, oracle/ctn_oracle.py): the loss, the per-tensor L2 norms of all 294
gradient tensors and every 997th element of the flat gradient (model.named_parameters() order) -- the yardstick of
tests/test_gpu_h3.py::test_paper_config_gradients_at_the_bench_batch.  Not reference-generated (the reference's loss is fp32-only,
SURVEY 8c): a yardstick for the arithmetics, pinned like tests/golden/paper_traj_fp64.npz.
Also writes benchmarks/_grad64_m<M>.pt (the full gradient rounded to fp32; git-ignored) for benchmarks/arith_grad_err.py.
usage: python oracle/make_grad_golden.py [M]      (M = 8: ~10 minutes on 8 cores)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import conv_tasnet_amd as ctn  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

STRIDE = 997
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
torch.manual_seed(0)
m = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C)
mix, lens, src = O.synth_batch(0, M, 32000)
sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in m.state_dict().items()}
t0 = time.time()
est = O.forward(cfg, sd, mix.double())
loss, max_snr, _, _ = O.cal_loss(src.double(), est, lens)
loss.backward()
names = [k for k, _ in m.named_parameters()]
g = torch.cat([sd[k].grad.reshape(-1) for k in names])
norms = np.asarray([float(sd[k].grad.norm()) for k in names])
print("fp64 oracle, M = %d: loss %.12f, |g| %.6e (%.0f s)" % (M, float(loss), float(g.norm()), time.time() - t0))
torch.save({"M": M, "loss": float(loss), "grad": g.float(), "names": names}, os.path.join(ROOT, "benchmarks", "_grad64_m%d.pt" % M))
if M == 8:
    np.savez(os.path.join(ROOT, "tests", "golden", "paper_grad_fp64_m8.npz"), loss=np.float64(float(loss)), g=g[::STRIDE].numpy(),
             norms=norms, gnorm=np.float64(float(g.norm())), stride=np.int64(STRIDE), M=np.int64(M),
             max_snr=max_snr.detach().numpy())
