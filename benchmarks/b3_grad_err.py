#!/usr/bin/env python
"""Per-parameter gradient error of the golden models under both GEMM arithmetics (max |g - ref| / max |ref|)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conftest import load_golden  # noqa: E402

DEV = "cuda:0"


def run(name, norm, arith):
    ctn.lib.call("ctn_tune", b"arith", arith)
    ops._ws_cache.clear()
    gd = load_golden(name)
    N, L, B, H, P, X, R, C = [int(v) for v in gd["cfg"]]
    if norm == "BN":
        m = ctn.ConvTasNet(N, L, B, H, P, X, R, C, norm_type=norm, causal=bool(int(gd["causal"])))
        m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in gd.items() if k.startswith("p0:")})
    else:
        m = ctn.ConvTasNet(N, L, B, H, P, X, R, C, norm_type=str(gd["norm_type"]), causal=bool(int(gd["causal"])),
                           mask_nonlinear=str(gd["mask_nonlinear"]))
        m.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in gd.items() if k.startswith("p:")})
    m = m.to(DEV).train()
    mix, src, lens = (torch.from_numpy(gd[k]).to(DEV) for k in ("mixture", "source", "lengths"))
    est = m(mix)
    est_err = float((est.double().cpu() - torch.from_numpy(gd["est_source_raw"]).double()).abs().max() / torch.from_numpy(gd["est_source_raw"]).abs().max())
    loss = ctn.cal_loss(src, est, lens)[0]
    loss.backward()
    errs = []
    for k, p in m.named_parameters():
        ref = torch.from_numpy(gd["g:" + k]).double()
        e = float((p.grad.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))
        errs.append((e, k, float(ref.abs().max())))
    errs.sort(reverse=True)
    print("%s arith=%d cfg=%s loss err %.2e dB, est err %.2e; worst gradients:" % (name, arith, (N, L, B, H, P, X, R, C), abs(float(loss.detach()) - float(gd["loss"])), est_err))
    for e, k, mx in errs[:3]:
        print("   %.3e  %-48s max|ref| %.3e" % (e, k, mx))


for name, norm in (("model_tiny_gln", "gLN"), ("model_tiny_cln_causal", "cLN"), ("model_c3_softmax", "gLN"), ("model_c3_relu_x4", "gLN")):
    for arith in (0, 1):
        run(name, norm, arith)
