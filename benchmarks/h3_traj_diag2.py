import os, sys
sys.path.insert(0, os.getcwd())
import torch
import conv_tasnet_amd as ctn
from conv_tasnet_amd.optim import FlatAdam
from conv_tasnet_amd.train import SyntheticLoader
DEV = "cuda:0"
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
def run(arith, steps):
    ctn.set_gemm_arith(arith)
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    snaps, gsn = [], []
    for _ in range(steps):
        opt.zero_grad()
        loss = ctn.cal_loss(src, m(mix), lens)[0]
        loss.backward()
        torch.cuda.synchronize()
        gsn.append({k: p.grad.detach().double().clone() for k, p in m.named_parameters()})
        opt.step(max_grad_norm=5.0)
        snaps.append({k: p.detach().double().clone() for k, p in m.named_parameters()})
    return snaps, gsn
s32, g32 = run("fp32", 3)
for a in ("b6", "h3"):
    s, g = run(a, 3)
    for step in range(3):
        rows = sorted(((float((s[step][k] - s32[step][k]).norm()), k, float((s[step][k] - s32[step][k]).abs().max())) for k in s[step]), reverse=True)
        tot = sum(r[0] ** 2 for r in rows) ** 0.5
        grow = sorted(((float((g[step][k] - g32[step][k]).norm() / (g32[step][k].norm() + 1e-300)), k) for k in g[step]), reverse=True)
        gt = (sum(float(((g[step][k] - g32[step][k]) ** 2).sum()) for k in g[step]) / sum(float((g32[step][k] ** 2).sum()) for k in g[step])) ** 0.5
        print("%s after step %d: |p - p32| %.3e; top: %s" % (a, step + 1, tot, "; ".join("%.2e(max %.1e) %s" % (r[0], r[2], r[1].replace("separator.network.", "")) for r in rows[:4])))
        print("      gradient at step %d: total rel diff %.3e; top: %s" % (step + 1, gt, "; ".join("%.2e %s" % (r[0], r[1].replace("separator.network.", "")) for r in grow[:3])))
