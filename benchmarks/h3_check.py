#!/usr/bin/env python
"""h3 GEMM forms (two fp16 pieces per operand under tracked power-of-two scales, ctn_*_h3) against fp64 torch, next to the
b6 and fp32-MFMA kernels on the same data: max / rms error in units of the row-wise sum of |a||b| (the natural scale of a dot
product's rounding error) and time per launch -- on unit-scale data and on operands with gradient-like magnitudes, large
magnitudes, per-utterance scales 2^-40 .. 2^40 apart and heavy tails.  usage: python benchmarks/h3_check.py"""
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


def err(o, ref, scale):
    e = (o.double() - ref).abs() / scale
    return e.max().item(), e.pow(2).mean().sqrt().item()


def line(name, fns, ref, scale, time=True):
    out = "%-26s" % name
    for tag, fn in fns:
        o = fn()
        mx, rms = err(o, ref, scale)
        out += " | %s %s max %.2e rms %.2e" % (tag, ("%6.1f us" % timeit(fn)) if time else "", mx, rms)
    print(out, flush=True)


def run(tag, sx, sg, sw, heavy=False, per_m=False, time=True):
    """sx / sg: magnitudes of the forward activations / the gradients; sw: of the weights."""
    torch.manual_seed(0)
    xB = torch.randn(M, B, Kp, device=dev) * sx
    xH = torch.randn(M, H, Kp, device=dev) * sx
    gB = torch.randn(M, B, Kp, device=dev) * sg
    gH = torch.randn(M, H, Kp, device=dev) * sg
    if heavy:          # heavy tails: a few elements 1e4 times the rest
        for t in (xB, xH, gB, gH):
            t.mul_(torch.where(torch.rand_like(t) < 1e-5, 1e4, 1.0))
    if per_m:          # every utterance at its own magnitude
        f = torch.tensor([2.0 ** e for e in (-40, -20, -8, 0, 3, 12, 24, 40)], device=dev).view(M, 1, 1)
        xB, xH, gB, gH = xB * f, xH * f, gB * f.flip(0), gH * f.flip(0)
    for t in (xB, xH, gB, gH):
        t[..., K:] = 0
    w1 = torch.randn(H, B, device=dev) * sw
    w2 = torch.randn(B, H, device=dev) * sw
    a = torch.full((1,), 0.25, device=dev)
    g = torch.randn(1, H, 1, device=dev)
    b = torch.randn(1, H, 1, device=dev)
    print("---- %s" % tag, flush=True)
    arith(2)
    p1, p2 = ops.h3_pieces(w1, H, B, False), ops.h3_pieces(w2, B, H, False)          # forward operands
    q2, q1 = ops.h3_pieces(w2, H, B, True), ops.h3_pieces(w1, B, H, True)            # input-gradient operands
    axB, axH, agB, agH = ops.absmax_rows(xB), ops.absmax_rows(xH), ops.absmax_rows(gB), ops.absmax_rows(gH)
    gbm = ops.absmax_of(g.view(-1), b.view(-1))
    w1t = w1.t().contiguous()
    w2t = w2.t().contiguous()

    def with_arith(v, fn):
        def f():
            arith(v)
            return fn()
        return f

    # K1: h1 = W1 x (+ statistics)
    ref = torch.einsum("rc,mck->mrk", w1.double(), xB.double())
    sc = torch.einsum("rc,mck->mrk", w1.double().abs(), xB.double().abs()).clamp_min(1e-300)
    line("K1 stats", [("h3", lambda: ops.pw_gemm_h3(p1, xB, H, B, K, axB, epi_alpha=a)[0]),
                      ("b6", with_arith(2, lambda: ops.pw_gemm(w1t, xB, H, B, K, trans_w=True, epi_alpha=a)[0])),
                      ("fp32", with_arith(0, lambda: ops.pw_gemm(w1t, xB, H, B, K, trans_w=True, epi_alpha=a)[0]))], ref, sc, time)
    # B5: dx = W1^T dh1 + dy
    ref5 = torch.einsum("cr,mck->mrk", w1.double(), gH.double()) + gB.double()
    sc5 = (torch.einsum("cr,mck->mrk", w1.double().abs(), gH.double().abs()) + gB.double().abs()).clamp_min(1e-300)
    oam = torch.zeros(M, 64, dtype=torch.int32, device=dev)
    line("B5 dgrad + residual", [("h3", lambda: ops.pw_gemm_h3(q1, gH, B, H, K, agH, residual=gB, out_amax=oam)[0]),
                                 ("b6", with_arith(2, lambda: ops.pw_gemm(w1, gH, B, H, K, trans_w=True, residual=gB)[0])),
                                 ("fp32", with_arith(0, lambda: ops.pw_gemm(w1, gH, B, H, K, trans_w=True, residual=gB)[0]))], ref5, sc5, time)
    true_am = ref5.float().abs().amax((1, 2))
    got_am = oam.view(torch.float32).amax(1)
    print("   out_amax vs max |fp64 result|: max rel diff %.2e" % ((got_am - true_am).abs() / true_am).max().item())
    # B1: dn2 = W2^T dy (+ gLN backward sums)
    pre = torch.where(xH >= 0, xH, 0.25 * xH).double()
    cnt = H * K
    mean = pre[..., :K].sum((1, 2)) / cnt
    var = (pre[..., :K] ** 2).sum((1, 2)) / cnt - mean * mean
    rstd = 1.0 / torch.sqrt(var + 1e-8)
    ms = torch.stack([mean, rstd], -1).float().contiguous()
    refb1 = torch.einsum("cr,mck->mrk", w2.double(), gB.double())
    scb1 = torch.einsum("cr,mck->mrk", w2.double().abs(), gB.double().abs()).clamp_min(1e-300)
    line("B1 dgrad gLN sums", [("h3", lambda: ops.pw_dgrad_gln_h3(q2, gB, H, B, K, xH, g, a, ms, agB)[0]),
                               ("b6", with_arith(2, lambda: ops.pw_dgrad_gln(w2, gB, H, B, K, xH, g, a, ms)[0])),
                               ("fp32", with_arith(0, lambda: ops.pw_dgrad_gln(w2, gB, H, B, K, xH, g, a, ms)[0]))], refb1, scb1, time)
    # K3: out = W2 gLN(prelu(d)) + x
    st2 = torch.stack([pre[..., :K].sum((1, 2)), (pre[..., :K] ** 2).sum((1, 2))], -1).reshape(M, 1, 2).contiguous()
    nrm = g.double() * ((pre - mean[:, None, None]) * rstd[:, None, None]) + b.double()
    nrm[..., K:] = 0
    ref3 = torch.einsum("rc,mck->mrk", w2.double(), nrm) + xB.double()
    sc3 = (torch.einsum("rc,mck->mrk", w2.double().abs(), nrm.abs()) + xB.double().abs()).clamp_min(1e-300)
    line("K3 pro + residual", [("h3", lambda: ops.pw_gemm_h3(p2, xH, B, H, K, axH, pro=(st2, g, b, a), gbmax=gbm, residual=xB)[0]),
                               ("b6", with_arith(2, lambda: ops.pw_gemm(w2t, xH, B, H, K, trans_w=True, pro=(st2, g, b, a), residual=xB)[0])),
                               ("fp32", with_arith(0, lambda: ops.pw_gemm(w2t, xH, B, H, K, trans_w=True, pro=(st2, g, b, a), residual=xB)[0]))], ref3, sc3, time)
    # weight gradients
    refw = torch.einsum("mrk,mck->rc", gH.double(), xB.double())
    scw = torch.einsum("mrk,mck->rc", gH.double().abs(), xB.double().abs()).clamp_min(1e-300)
    line("B6 wgrad dW1", [("h3", lambda: ops.pw_wgrad_h3(gH, xB, H, B, K, agH, axB)),
                          ("b6", with_arith(2, lambda: ops.pw_wgrad(gH, xB, H, B, K))),
                          ("fp32", with_arith(0, lambda: ops.pw_wgrad(gH, xB, H, B, K)))], refw, scw, time)
    refw2 = torch.einsum("mrk,mck->rc", gB.double(), nrm)
    scw2 = torch.einsum("mrk,mck->rc", gB.double().abs(), nrm.abs()).clamp_min(1e-300)
    line("B2 wgrad dW2 pro", [("h3", lambda: ops.pw_wgrad_h3(gB, xH, B, H, K, agB, axH, pro=(g, b, a, ms), gbmax=gbm)),
                              ("b6", with_arith(2, lambda: ops.pw_wgrad(gB, xH, B, H, K, pro=(g, b, a, ms)))),
                              ("fp32", with_arith(0, lambda: ops.pw_wgrad(gB, xH, B, H, K, pro=(g, b, a, ms))))], refw2, scw2, time)


run("unit scale (activations 1, gradients 1, weights 0.05)", 1.0, 1.0, 0.05)
run("training-like (activations 3, gradients 1e-7, weights 0.05)", 3.0, 1e-7, 0.05, time=False)
run("tiny gradients 1e-20, weights 1e-3", 1.0, 1e-20, 1e-3, time=False)
run("large: activations 1e6, gradients 1e5, weights 30", 1e6, 1e5, 30.0, time=False)
run("heavy tails (1e-5 of the elements x 1e4)", 1.0, 1e-6, 0.05, heavy=True, time=False)
run("per-utterance magnitudes 2^-40 .. 2^40", 1.0, 1e-3, 0.05, per_m=True, time=False)
arith(3)
