#!/usr/bin/env python
"""How far do training trajectories drift apart under DIFFERENT but equally valid fp32 arithmetics?  The paper config is trained
for a few steps from the same weights on the same batch by (a) the CPU oracle -- PyTorch fp32 on the host, the reference's own
arithmetic -- and (b) the HIP path under the fp32 MFMA, b6 and h3.  Prints per-step losses and the distance of every run's final
parameters from the CPU oracle's, in units of the distance travelled.  usage: python benchmarks/trajectory_vs_oracle.py [steps] [M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

DEV = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
mix, lens, src = O.synth_batch(0, M, 32000)
torch.manual_seed(0)
m0 = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C)
init = {k: v.detach().clone() for k, v in m0.state_dict().items()}
names = [k for k, _ in m0.named_parameters()]

torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
sd = {k: v.clone() for k, v in init.items()}
st, lo = {}, []
for i in range(steps):
    lo.append(O.train_step(cfg, sd, st, mix, src, lens))
    print("oracle step %d loss %.6f" % (i, lo[-1]), flush=True)
runs = {"cpu fp32 (oracle)": (lo, torch.cat([sd[k].reshape(-1) for k in names]))}
p0 = torch.cat([init[k].reshape(-1) for k in names])
for arith in ("fp32", "b6", "h3"):
    ctn.set_gemm_arith(arith)
    m = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C)
    m.load_state_dict(init)
    m = m.to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    ls = []
    for _ in range(steps):
        opt.zero_grad()
        loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
        loss.backward()
        opt.step(max_grad_norm=5.0)
        ls.append(float(loss.detach()))
    runs["hip " + arith] = (ls, torch.cat([p.detach().reshape(-1).cpu() for _, p in m.named_parameters()]))
ctn.set_gemm_arith("h3")
ref_l, ref_p = runs["cpu fp32 (oracle)"]
trav = float((ref_p - p0).double().norm())
print("paper config, M = %d, %d steps; distance travelled by the parameters %.4e" % (M, steps, trav))
for k, (ls, p) in runs.items():
    print("%-18s losses %s" % (k, " ".join("%.6f" % v for v in ls)))
for k, (ls, p) in runs.items():
    print("%-18s max |loss - oracle| %.2e dB   |p - p_oracle| / travelled %.3e   |p - p_hip_fp32| / travelled %.3e"
          % (k, max(abs(a - b) for a, b in zip(ls, ref_l)), float((p - ref_p).double().norm()) / trav,
             float((p - runs["hip fp32"][1]).double().norm()) / trav))
