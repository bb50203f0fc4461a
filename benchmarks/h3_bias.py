"""Is the error of a GEMM arithmetic biased?  Positive operands: mean SIGNED relative error of every form against fp64."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import conv_tasnet_amd as ctn
from conv_tasnet_amd import ops
dev = "cuda:0"
M, K, B, H = 4, 3199, 256, 512
Kp = ops.padded_frames(K)
torch.manual_seed(0)
for tag, f in (("positive operands", lambda t: t.abs()), ("signed operands", lambda t: t)):
    x = f(torch.randn(M, B, Kp, device=dev)); x[..., K:] = 0
    g = f(torch.randn(M, H, Kp, device=dev)); g[..., K:] = 0
    w = f(torch.randn(H, B, device=dev)) * 0.05
    ref = torch.einsum("rc,mck->mrk", w.double(), x.double())[..., :K]
    refw = torch.einsum("mrk,mck->rc", g.double(), x.double())
    sc = torch.einsum("rc,mck->mrk", w.double().abs(), x.double().abs())[..., :K]
    scw = torch.einsum("mrk,mck->rc", g.double().abs(), x.double().abs())
    ax, ag = ops.absmax_rows(x), ops.absmax_rows(g)
    outs = {"h3": (ops.pw_gemm_h3(ops.h3_pieces(w, H, B, False), x, H, B, K, ax)[0], ops.pw_wgrad_h3(g, x, H, B, K, ag, ax))}
    for a in ("b6", "fp32"):
        with ctn.gemm_arithmetic(a):
            outs[a] = (ops.pw_gemm(w, x, H, B, K)[0], ops.pw_wgrad(g, x, H, B, K))
    outs["torch fp32 (rocBLAS)"] = (torch.einsum("rc,mck->mrk", w, x), torch.einsum("mrk,mck->rc", g, x))
    for a, (o, ow) in outs.items():
        e = (o[..., :K].double() - ref) / sc
        ew = (ow.double() - refw) / scw
        print("%-18s %-22s fwd: mean %+.3e rms %.3e | wgrad: mean %+.3e rms %.3e" % (tag, a, e.mean().item(), e.pow(2).mean().sqrt().item(), ew.mean().item(), ew.pow(2).mean().sqrt().item()), flush=True)
