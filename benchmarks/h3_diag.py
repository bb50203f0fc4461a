import os, sys
sys.path.insert(0, os.getcwd())
import torch
import conv_tasnet_amd as ctn
from oracle import ctn_oracle as O
DEV = "cuda:0"
cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
torch.manual_seed(0)
m = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C).to(DEV)
from conv_tasnet_amd.train import SyntheticLoader
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
def grads(arith, sel=None):
    ctn.set_gemm_arith(arith)
    m.zero_grad()
    mx, ln, sr = (mix, lens, src) if sel is None else (mix[sel], lens[sel], src[sel])
    loss = ctn.cal_loss(sr, m(mx), ln)[0]
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.detach().double().clone() for k, p in m.named_parameters()}
l32, g32 = grads("fp32")
for arith in ("b6", "h3"):
    l, g = grads(arith)
    rows = sorted(((float((g[k] - g32[k]).norm() / (g32[k].norm() + 1e-300)), k, float(g32[k].norm())) for k in g), reverse=True)
    tot = (sum(float(((g[k] - g32[k]) ** 2).sum()) for k in g) / sum(float((g32[k] ** 2).sum()) for k in g)) ** 0.5
    print(arith, "loss diff %.2e, total rel L2 diff vs fp32 %.3e; worst tensors:" % (abs(l - l32), tot))
    for r in rows[:8]:
        print("    %.3e %-50s |g| %.3e" % r)
# per-utterance: which utterance makes h3 deviate?
for u in range(8):
    sel = slice(u, u + 1)
    l32u, g32u = grads("fp32", sel)
    lh, gh = grads("h3", sel)
    tot = (sum(float(((gh[k] - g32u[k]) ** 2).sum()) for k in gh) / sum(float((g32u[k] ** 2).sum()) for k in gh)) ** 0.5
    worst = max((float((gh[k] - g32u[k]).norm() / (g32u[k].norm() + 1e-300)), k) for k in gh)
    print("utt %d: h3 vs fp32 total %.3e worst %.3e %s" % (u, tot, worst[0], worst[1]))
