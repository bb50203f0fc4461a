#!/usr/bin/env python
"""Paper config, one training step's gradients under every GEMM arithmetic against the fp64 CPU oracle on the same weights and
data: loss error, relative L2 error over all parameters, worst per-tensor error.  usage: python benchmarks/arith_grad_err.py [M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
torch.manual_seed(0)
m = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C).to(DEV)
mix, lens, src = O.synth_batch(0, M, 32000)
torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in m.state_dict().items()}
est_ref = O.forward(cfg, sd, mix.double())
loss_ref = O.cal_loss(src.double(), est_ref, lens)[0]
loss_ref.backward()
ref = {k: v.grad for k, v in sd.items() if v.grad is not None}
tot = sum(float((g ** 2).sum()) for g in ref.values()) ** 0.5
print("fp64 oracle: loss %.9f, |g| %.4e" % (float(loss_ref), tot), flush=True)
for arith in ("h3", "b6", "fp32"):
    ctn.set_gemm_arith(arith)
    m.zero_grad()
    est = m(mix.to(DEV))
    loss = ctn.cal_loss(src.to(DEV), est, lens.to(DEV))[0]
    loss.backward()
    e_est = float((est.detach().double().cpu() - est_ref.detach()).abs().max() / est_ref.detach().abs().max())
    err2, rows = 0.0, []
    for k, p in m.named_parameters():
        d = p.grad.detach().double().cpu() - ref[k]
        err2 += float((d ** 2).sum())
        rows.append((float(d.abs().max() / (ref[k].abs().max() + 1e-300)), float(d.norm() / (ref[k].norm() + 1e-300)), k))
    rows.sort(reverse=True)
    print("%-4s loss err %.2e dB, waveform err %.2e, gradient |g - g64| / |g64| = %.3e, worst tensor (max-norm) %.2e %s, median tensor L2 err %.2e"
          % (arith, abs(float(loss.detach()) - float(loss_ref)), e_est, err2 ** 0.5 / tot, rows[0][0], rows[0][2],
             sorted(r[1] for r in rows)[len(rows) // 2]), flush=True)
    by_kind = {}
    for mx, l2, k in rows:
        kind = k.split(".")[-2] + "." + k.split(".")[-1] if "network" in k else k
        by_kind.setdefault(kind, []).append(l2)
ctn.set_gemm_arith("h3")
