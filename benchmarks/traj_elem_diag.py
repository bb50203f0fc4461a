#!/usr/bin/env python
"""Follow-up of traj_adam_diag.py: at step 5 of the M = 8 paper-config run almost all of |update_h3 - update_fp32| sits in the mask
1x1 convolution's weight (separator.network.3.weight) in elements with small gradients.  Print the gradient / update history of the
elements that differ most, under the three arithmetics, and the distribution of |g| in that tensor.
usage: python benchmarks/traj_elem_diag.py [step] [M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

DEV = "cuda:0"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 5
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mix, lens, src = O.synth_batch(0, M, 32000)
mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
hist = {}
for arith in ("fp32", "b6", "h3"):
    ctn.set_gemm_arith(arith)
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    lay = {k: (o, q.numel(), tuple(q.shape)) for (k, q), o in zip(m.named_parameters(), opt._offsets)}
    o, n, shp = lay["separator.network.3.weight"]
    G, U, Mo, V, P = [], [], [], [], []
    for s in range(S + 2):
        opt.zero_grad()
        loss = ctn.cal_loss(src, m(mix), lens)[0]
        loss.backward()
        ctn.ops.join_side_stream(opt.flat_grads.device)
        torch.cuda.synchronize()
        G.append(opt.flat_grads[o:o + n].clone())
        p0 = opt.flat_params[o:o + n].clone()
        P.append(p0)
        opt.step(max_grad_norm=5.0)
        torch.cuda.synchronize()
        U.append(opt.flat_params[o:o + n] - p0)
        Mo.append(opt.exp_avg[o:o + n].clone())
        V.append(opt.exp_avg_sq[o:o + n].clone())
    hist[arith] = dict(G=G, U=U, Mo=Mo, V=V, P=P, norm=float(opt.last_total_norm))
ctn.set_gemm_arith("h3")
print("separator.network.3.weight %s; total gradient norm at the last step %.3f (clip at 5)" % (shp, hist["fp32"]["norm"]))
for s in range(S + 2):
    g = hist["fp32"]["G"][s].abs()
    qs = torch.quantile(g.float().cpu(), torch.tensor([0.01, 0.1, 0.5, 0.9, 0.99]))
    dzero = int((g == 0).sum())
    print("step %d: |g| quantiles 1/10/50/90/99 %% = %s ; exact zeros %d ; max %.2e ; |upd_h3 - upd_fp32| %.3e |upd_b6 - upd_fp32| %.3e ; |g_h3 - g_fp32| %.3e |g_b6 - g_fp32| %.3e" %
          (s, " ".join("%.1e" % v for v in qs.tolist()), dzero, float(g.max()),
           float((hist["h3"]["U"][s] - hist["fp32"]["U"][s]).norm()), float((hist["b6"]["U"][s] - hist["fp32"]["U"][s]).norm()),
           float((hist["h3"]["G"][s] - hist["fp32"]["G"][s]).norm()), float((hist["b6"]["G"][s] - hist["fp32"]["G"][s]).norm())))
du = (hist["h3"]["U"][S] - hist["fp32"]["U"][S]).abs()
top = du.topk(6).indices.tolist()
rows = sorted(set(i // shp[1] for i in du.topk(2000).indices.tolist()))
print("step %d: the 2000 most different elements sit in %d of %d output rows: %s" % (S, len(rows), shp[0], rows[:40]))
for i in top:
    print("element %d (row %d, col %d):" % (i, i // shp[1], i % shp[1]))
    for a in ("fp32", "b6", "h3"):
        print("   %-4s g    %s" % (a, " ".join("%+.3e" % float(hist[a]["G"][s][i]) for s in range(S + 2))))
        print("   %-4s upd  %s" % (a, " ".join("%+.3e" % float(hist[a]["U"][s][i]) for s in range(S + 2))))
# row-level view: gradient norm of the most affected rows per step and arithmetic
for r in rows[:4]:
    print("row %d: |g_row| per step" % r)
    for a in ("fp32", "b6", "h3"):
        print("   %-4s %s" % (a, " ".join("%.3e" % float(hist[a]["G"][s].view(shp[0], shp[1])[r].norm()) for s in range(S + 2))))
