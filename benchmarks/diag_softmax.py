import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import conv_tasnet_amd as ctn
from oracle import ctn_oracle as O
cfg = O.Config(N=256, L=16, B=256, H=512, P=3, X=8, R=2, C=3, mask_nonlinear="softmax")
torch.manual_seed(5)
m = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C, mask_nonlinear="softmax").cuda()
mix, lens, src = O.synth_batch(50, 2, 32000, C=3, sr=16000)
lens = lens.clone(); lens[-1] = 32000 - 1234; mix[-1, lens[-1]:] = 0; src[-1, :, lens[-1]:] = 0
sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in m.state_dict().items()}
est_ref = O.forward(cfg, sd, mix.double())
loss_ref = O.cal_loss(src.double(), est_ref, lens)[0]
loss_ref.backward()
sd32 = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
l32 = O.cal_loss(src, O.forward(cfg, sd32, mix), lens)[0]
l32.backward()
est = m(mix.cuda())
loss = ctn.cal_loss(src.cuda(), est, lens.cuda())[0]
loss.backward()
print("loss", float(loss), float(loss_ref), float(l32))
rows = []
for k, p in m.named_parameters():
    r = sd[k].grad
    e_hip = float((p.grad.cpu().double() - r).abs().max() / (r.abs().max() + 1e-300))
    e_cpu = float((sd32[k].grad.double() - r).abs().max() / (r.abs().max() + 1e-300))
    rows.append((e_hip, e_cpu, float(r.abs().max()), k))
rows.sort(reverse=True)
for r in rows[:12]:
    print("hip %.2e  cpu32 %.2e  |g| %.2e  %s" % r)
