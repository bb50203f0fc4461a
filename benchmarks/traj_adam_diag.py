#!/usr/bin/env python
"""Which part of the training step turns the 1e-5-level gradient differences between GEMM arithmetics into 4e-3 dB of loss after
six steps (profiles/r03_h3_traj_vs_cpu_oracle_m8.txt)?  Paper config, bench batch, three optimiser rules from the same weights:
  adam      the reference step: clip(5) + Adam(lr 1e-3, eps 1e-8)                       (src/solver.py:194-196)
  adam_eps  the same with eps = 1e-4 (elements with |g| << 1e-4 take proportionally small steps)
  ngd       normalised gradient descent p -= eta g / |g| (linear in the gradient direction, no per-element division)
For each rule: max |loss_h3 - loss_fp32| and |loss_b6 - loss_fp32| over the run; for adam also WHERE the updates of h3 and fp32
differ: per step the share of |update_h3 - update_fp32|^2 that sits in elements whose fp32 gradient is below 1e-7 / 1e-6 / 1e-5 of
the largest gradient element, and the tensors that carry most of it.
usage: python benchmarks/traj_adam_diag.py [steps] [M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from oracle import ctn_oracle as O  # noqa: E402

DEV = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mix, lens, src = O.synth_batch(0, M, 32000)
mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)


def run(arith, rule):
    ctn.set_gemm_arith(arith)
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3, eps=1e-4 if rule == "adam_eps" else 1e-8)
    rec = []
    for s in range(steps):
        opt.zero_grad()
        loss = ctn.cal_loss(src, m(mix), lens)[0]
        loss.backward()
        ctn.ops.join_side_stream(opt.flat_grads.device)
        torch.cuda.synchronize()
        g = opt.flat_grads.clone()
        p0 = opt.flat_params.clone()
        if rule == "ngd":
            with torch.no_grad():
                opt.flat_params.add_(g, alpha=-1.0 / float(g.double().norm()))
        else:
            opt.step(max_grad_norm=5.0)
        torch.cuda.synchronize()
        rec.append(dict(loss=float(loss.detach()), g=g, upd=opt.flat_params - p0))
    names, offs, sizes = [], [], []
    for (k, q), o in zip(m.named_parameters(), opt._offsets):
        names.append(k); offs.append(o); sizes.append(q.numel())
    return rec, (names, offs, sizes)


for rule in ("adam", "adam_eps", "ngd"):
    runs = {}
    for arith in ("fp32", "b6", "h3"):
        runs[arith], layout = run(arith, rule)
    names, offs, sizes = layout
    print("== rule %s" % rule)
    for a in ("fp32", "b6", "h3"):
        print("   %-4s losses %s" % (a, " ".join("%.6f" % r["loss"] for r in runs[a])))
    for a in ("b6", "h3"):
        print("   max |loss_%s - loss_fp32| = %.2e dB" % (a, max(abs(x["loss"] - y["loss"]) for x, y in zip(runs[a], runs["fp32"]))))
    if rule != "adam":
        continue
    for s in range(steps):
        g32 = runs["fp32"][s]["g"]
        gmax = float(g32.abs().max())
        for a in ("b6", "h3"):
            du = (runs[a][s]["upd"] - runs["fp32"][s]["upd"]).double()
            tot = float((du ** 2).sum())
            shares = []
            for thr in (1e-7, 1e-6, 1e-5, 1e-4):
                sel = g32.abs() < thr * gmax
                shares.append("|g|<%.0e gmax: %4.1f%% of d_upd^2 in %d elements" % (thr, 100 * float((du[sel] ** 2).sum()) / max(tot, 1e-300), int(sel.sum())))
            per = sorted(((float((du[o:o + n] ** 2).sum()), k, n) for k, o, n in zip(names, offs, sizes)), reverse=True)[:3]
            print("   step %d %-3s-fp32: |d_upd| %.3e (|upd| %.3e); %s; top tensors: %s" %
                  (s, a, tot ** 0.5, float(runs["fp32"][s]["upd"].double().norm()), "; ".join(shares),
                   ", ".join("%s[%d] %.0f%%" % (k.replace("separator.network.", ""), n, 100 * v / max(tot, 1e-300)) for v, k, n in per)))
ctn.set_gemm_arith("h3")
