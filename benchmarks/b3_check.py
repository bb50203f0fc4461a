#!/usr/bin/env python
"""b3 (split-bf16) GEMM forms against fp64 torch and against the fp32-MFMA kernels: max error relative to the row-wise
sum of |a||b| (the natural scale of a dot product's rounding error), and time per launch.  usage: python benchmarks/b3_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

dev = "cuda:0"
M, K = 8, 3199
Kp = ops.padded_frames(K)
B, H = 256, 512
torch.manual_seed(0)
xB = torch.randn(M, B, Kp, device=dev); xB[..., K:] = 0
xH = torch.randn(M, H, Kp, device=dev); xH[..., K:] = 0
w1 = torch.randn(H, B, device=dev) * 0.05
w2 = torch.randn(B, H, device=dev) * 0.05
a = torch.full((1,), 0.25, device=dev)
g = torch.randn(1, H, 1, device=dev)
b = torch.randn(1, H, 1, device=dev)
D = torch.randn(H, 1, 3, device=dev)
ms = torch.tensor([[0.1, 1.3]] * M, device=dev)


def arith(v):
    ctn.lib.call("ctn_tune", b"arith", v)
    ops._ws_cache.clear()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def report(name, fn, ref, scale):
    """max and rms error against fp64, in units of sum |a||b| (the natural scale of a dot product's rounding error)."""
    line = "%-24s" % name
    for v, tag in ((0, "fp32"), (2, "b6"), (1, "b3")):
        arith(v)
        o = fn()
        t = timeit(fn)
        e = (o.double() - ref).abs() / scale
        line += " | %s %6.1f us max %.2e rms %.2e" % (tag, t, e.max().item(), e.pow(2).mean().sqrt().item())
    print(line, flush=True)


pre = torch.where(xH >= 0, xH, 0.25 * xH).double()
st2 = torch.stack([pre[..., :K].sum((1, 2)), (pre[..., :K] ** 2).sum((1, 2))], -1).reshape(M, 1, 2).contiguous()   # [M, 1 part, (sum, sumsq)]
# plain forward, both weight layouts
ref = torch.einsum("rc,mck->mrk", w1.double(), xB.double())
sc = torch.einsum("rc,mck->mrk", w1.double().abs(), xB.double().abs()).clamp_min(1e-30)
report("plain [R,Cn]", lambda: ops.pw_gemm(w1, xB, H, B, K)[0], ref, sc)
w1t = w1.t().contiguous()
report("plain W^T", lambda: ops.pw_gemm(w1t, xB, H, B, K, trans_w=True)[0], ref, sc)
report("K1 stats W^T", lambda: ops.pw_gemm(w1t, xB, H, B, K, trans_w=True, epi_alpha=a)[0], ref, sc)
# residual + dgrad form
ref5 = torch.einsum("cr,mck->mrk", w1.double(), xH.double()) + xB.double()
sc5 = torch.einsum("cr,mck->mrk", w1.double().abs(), xH.double().abs()) + xB.double().abs()
report("B5 dgrad + residual", lambda: ops.pw_gemm(w1, xH, B, H, K, trans_w=True, residual=xB)[0], ref5, sc5.clamp_min(1e-30))
refb1 = torch.einsum("cr,mck->mrk", w2.double(), xB.double())
scb1 = torch.einsum("cr,mck->mrk", w2.double().abs(), xB.double().abs()).clamp_min(1e-30)
report("B1 dgrad gLN sums", lambda: ops.pw_dgrad_gln(w2, xB, H, B, K, xH, g, a, ms)[0], refb1, scb1)
# K3: prologue + residual: reference through the fp32 kernel's own normalised operand (computed in fp64 here)
arith(0)
part = st2.double().sum(1)
cnt = H * K
mean = part[:, 0] / cnt
var = part[:, 1] / cnt - mean * mean
rstd = 1.0 / torch.sqrt(var + 1e-8)
nrm = g.double() * ((pre - mean[:, None, None]) * rstd[:, None, None]) + b.double()
nrm[..., K:] = 0
w2t = w2.t().contiguous()
ref3 = torch.einsum("rc,mck->mrk", w2.double(), nrm) + xB.double()
sc3 = torch.einsum("rc,mck->mrk", w2.double().abs(), nrm.abs()) + xB.double().abs()
report("K3 pro + residual W^T", lambda: ops.pw_gemm(w2t, xH, B, H, K, trans_w=True, pro=(st2, g, b, a), residual=xB)[0], ref3, sc3.clamp_min(1e-30))
# weight gradients
refw = torch.einsum("mrk,mck->rc", xH.double(), xB.double())
scw = torch.einsum("mrk,mck->rc", xH.double().abs(), xB.double().abs()).clamp_min(1e-30)
report("wgrad dW1", lambda: ops.pw_wgrad(xH, xB, H, B, K), refw, scw)
nrm2 = g.double() * ((pre - 0.1) * 1.3) + b.double()
nrm2[..., K:] = 0
refw2 = torch.einsum("mrk,mck->rc", xB.double(), nrm2)
scw2 = torch.einsum("mrk,mck->rc", xB.double().abs(), nrm2.abs()).clamp_min(1e-30)
report("wgrad dW2 pro", lambda: ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms)), refw2, scw2)
from conv_tasnet_amd.ops import _p, _b3_pieces  # noqa: E402
outH = torch.empty(M, H, Kp, device=dev)
outB = torch.empty(M, B, Kp, device=dev)
for av, tile in [(a_, t_) for a_ in (2, 1) for t_ in (1, 2, 0)]:      # 128x64, 256x64, 128x128: the kernels on pre-split weights (the composite's forms), GEMM alone
    ctn.lib.call("ctn_tune", b"b3_tile", tile)
    ops._ws_cache.clear()
    arith(av)
    p1, p2 = _b3_pieces(w1, H, B, False), _b3_pieces(w2, B, H, False)          # forward operands
    q2, q1 = _b3_pieces(w2, H, B, True), _b3_pieces(w1, B, H, True)            # input-gradient operands
    part = torch.empty((M, ctn.lib.ctn_pw_stats_parts(M, H, Kp), 2), dtype=torch.float64, device=dev)
    sm = ops._stream()
    k1 = lambda: ctn.lib.call("ctn_pw_gemm", _p(p1), _p(xB), _p(outH), M, H, B, K, Kp, 2, None, 0, None, None, None, None, None, _p(a), _p(part), 0, sm)
    k3 = lambda: ctn.lib.call("ctn_pw_gemm", _p(p2), _p(xH), _p(outB), M, B, H, K, Kp, 2, _p(st2), 1, _p(g), _p(b), _p(a), None, _p(xB), None, None, 0, sm)
    b1 = lambda: ctn.lib.call("ctn_pw_dgrad_gln_planes", _p(q2), _p(xB), _p(outH), M, H, B, K, Kp, _p(xH), _p(g), _p(a), _p(ms), _p(part), sm)
    b5 = lambda: ctn.lib.call("ctn_pw_gemm", _p(q1), _p(xH), _p(outB), M, B, H, K, Kp, 2, None, 0, None, None, None, None, _p(xB), None, None, 0, sm)
    print(("b6" if av == 2 else "b3") + " b3_tile=%d (pre-split weights)  K1 %6.1f  K3 %6.1f  B1 %6.1f  B5 %6.1f us" % (tile, timeit(k1), timeit(k3), timeit(b1), timeit(b5)), flush=True)
ctn.lib.call("ctn_tune", b"b3_tile", 1)
arith(2)
