import os, sys
sys.path.insert(0, os.getcwd())
import torch
import conv_tasnet_amd as ctn
from oracle import ctn_oracle as O
DEV = "cuda:0"
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
torch.manual_seed(0)
m = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C).to(DEV)
from conv_tasnet_amd.train import SyntheticLoader
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix[U:U + 1], lens[U:U + 1], src[U:U + 1]
torch.set_num_threads(16)
sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in m.state_dict().items()}
loss_ref = O.cal_loss(src.double(), O.forward(cfg, sd, mix.double()), lens)[0]
loss_ref.backward()
ref = {k: v.grad for k, v in sd.items() if v.grad is not None}
tot = sum(float((g ** 2).sum()) for g in ref.values()) ** 0.5
res = {}
for arith in ("h3", "b6", "fp32"):
    ctn.set_gemm_arith(arith)
    m.zero_grad()
    loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
    loss.backward()
    torch.cuda.synchronize()
    res[arith] = {k: float((p.grad.double().cpu() - ref[k]).norm()) for k, p in m.named_parameters()}
    print(arith, "total |g - g64| / |g64| = %.3e" % (sum(v ** 2 for v in res[arith].values()) ** 0.5 / tot))
rows = sorted(((res["h3"][k], res["b6"][k], res["fp32"][k], float(ref[k].norm()), k) for k in res["h3"]), reverse=True)
print("largest ABSOLUTE h3 errors:  h3 / b6 / fp32 / |g64| / name")
for r in rows[:12]:
    print("   %.3e %.3e %.3e  %.3e %s" % r)
