#!/usr/bin/env python
"""Why do 10-step trajectories of the paper config under different GEMM arithmetics separate (profiles/r03_h3_traj_*)?  Every
arithmetic trains from the same weights on the bench's batch; per step this prints, per utterance, the PIT decision (index of the
best permutation) and its MARGIN (SI-SNR of the best minus the second-best permutation, evaluated in fp64 on the stored estimate),
and between arithmetics the distance of the parameter vectors and of the gradients, per parameter family.
usage: python benchmarks/traj_perm_diag.py [steps] [M]"""
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


def sisnr_all_perms(src, est):
    """[M, C!] SI-SNR of every permutation in fp64 (zero-mean, as src/pit_criterion.py:40-77)"""
    s, e = src.double(), est.double()
    s = s - s.mean(-1, keepdim=True)
    e = e - e.mean(-1, keepdim=True)
    dot = torch.einsum("mit,mjt->mij", e, s)                      # [M, est i, src j]
    es = (s * s).sum(-1)                                          # [M, j]
    proj_e = dot ** 2 / (es[:, None, :] + 1e-8)                   # |s_target|^2 of (i, j)
    ee = (e * e).sum(-1)                                          # [M, i]
    noise = ee[:, :, None] - 2 * dot ** 2 / (es[:, None, :] + 1e-8) + proj_e
    snr = 10 * torch.log10(proj_e / (noise + 1e-8) + 1e-8)        # [M, i, j]
    C = s.shape[1]
    import itertools
    out = []
    for p in itertools.permutations(range(C)):
        out.append(sum(snr[:, p[j], j] for j in range(C)) / C)    # estimate p[j] explains source j
    return torch.stack(out, 1)


def family(name):
    p = name.split(".")
    if "network" in name and len(p) >= 3:
        return ".".join(p[-2:]) if p[-2].isdigit() is False else p[-1]
    return name


runs = {}
for arith in ("fp32", "b6", "h3"):
    ctn.set_gemm_arith(arith)
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
    names = [k for k, _ in m.named_parameters()]
    opt = FlatAdam(m.parameters(), lr=1e-3)
    rec = []
    for s in range(steps):
        opt.zero_grad()
        est = m(mix)
        loss, max_snr, est_m, _ = ctn.cal_loss(src, est, lens)
        snr = sisnr_all_perms(src, est_m.detach())
        srt = snr.sort(1, descending=True).values
        loss.backward()
        g = torch.cat([p.grad.detach().reshape(-1) for _, p in m.named_parameters()]).clone()
        opt.step(max_grad_norm=5.0)
        p = torch.cat([q.detach().reshape(-1) for _, q in m.named_parameters()]).clone()
        rec.append(dict(loss=float(loss.detach()), best=snr.argmax(1).cpu(), margin=(srt[:, 0] - srt[:, 1]).cpu(), g=g, p=p,
                        per_utt=snr.max(1).values.cpu()))
    runs[arith] = rec
    sizes = [q.numel() for _, q in m.named_parameters()]
ctn.set_gemm_arith("h3")
print("paper config, M = %d, %d steps" % (M, steps))
for s in range(steps):
    print("step %d: loss fp32 %.6f b6 %.6f h3 %.6f" % (s, runs["fp32"][s]["loss"], runs["b6"][s]["loss"], runs["h3"][s]["loss"]))
    for a in ("fp32", "b6", "h3"):
        r = runs[a][s]
        print("   %-4s best perm %s   margin [dB] %s" % (a, r["best"].tolist(), " ".join("%.2e" % v for v in r["margin"].tolist())))
    for a in ("b6", "h3"):
        dg = (runs[a][s]["g"] - runs["fp32"][s]["g"]).double()
        dp = (runs[a][s]["p"] - runs["fp32"][s]["p"]).double()
        print("   %-4s vs fp32: |dg| / |g| = %.3e   |dp| = %.3e   per-utterance SI-SNR difference [dB] %s" %
              (a, float(dg.norm() / runs["fp32"][s]["g"].double().norm()), float(dp.norm()),
               " ".join("%+.1e" % v for v in (runs[a][s]["per_utt"] - runs["fp32"][s]["per_utt"]).tolist())))
    # which families carry the h3 - fp32 gradient difference at this step
    off, fam = 0, {}
    dg = (runs["h3"][s]["g"] - runs["fp32"][s]["g"]).double()
    gg = runs["fp32"][s]["g"].double()
    for n, k in zip(names, sizes):
        f = n.split(".")[-1] if "network" in n else n
        if "network" in n:
            parts = n.split(".")
            f = parts[-2] + "." + parts[-1] if not parts[-2].isdigit() else parts[-1]
        a_, b_ = fam.get(f, (0.0, 0.0))
        fam[f] = (a_ + float((dg[off:off + k] ** 2).sum()), b_ + float((gg[off:off + k] ** 2).sum()))
        off += k
    top = sorted(fam.items(), key=lambda kv: -kv[1][0])[:4]
    print("   h3 - fp32 gradient difference by family (|d|^2 share, relative): " +
          ", ".join("%s %.0f%% (%.1e)" % (f, 100 * a_ / float((dg ** 2).sum()), (a_ / max(b_, 1e-300)) ** 0.5) for f, (a_, b_) in top))
