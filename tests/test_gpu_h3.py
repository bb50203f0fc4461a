"""GPU: the "h3" GEMM arithmetic of the composite stacks (include/ctn_hip.h, "h3" section; csrc/ctn_gemm_b3.h) through its own
entry points -- two fp16 pieces per fp32 operand under power-of-two scales derived from TRACKED maxima, three f16 MFMAs.

What is asserted, with the limits of the fp32 arithmetic (nothing is widened for h3):
  * every GEMM form (K1 statistics, K3 prologue + residual, B1 gLN-backward sums, B5 residual, both weight gradients) against
    fp64 on unit-scale data AND on operands whose magnitudes the 5-bit exponent of fp16 could not hold unscaled: gradient-like
    1e-20, large 1e6, utterances 2^-40 .. 2^40 apart, heavy tails -- error <= 6e-7 of sum |a||b| (the limit tests/test_gpu_b3.py
    applies to the default arithmetic) or, where the fp32 MFMA itself is above that (heavy tails), <= 1.25x the fp32 MFMA's error
    on the same data.  Observed on the MI355X (benchmarks/h3_check.py): 2.4e-7 against 3.9e-7 (b6) and 3.8e-7 (fp32 MFMA);
  * the tracked maxima are EXACT (ctn_absmax_rows, residual epilogue, ctn_dw_fwd, ctn_gln_prelu_bwd, weight maxima), so the
    per-kernel path that measures them and the composite that tracks them agree bit for bit (tests/test_gpu_train.py);
  * edge cases: an all-zero utterance, NaN / inf confined to their utterance, elements far below their utterance's maximum
    (absolute error <= 2^-38 of the maximum: graceful, documented loss of RELATIVE precision);
  * at the paper config: the gradients of one training step against the fp64 CPU oracle at M = 1 (computed in the test) and at
    the bench's batch M = 8 (fixture) -- h3's error is not above the fp32 MFMA's (observed 2.7e-6 / 3.1e-6 against 1.5e-5 /
    1.7e-5 and 1.8e-5 / 2.0e-5 for b6) -- and a 10-step trajectory against the same 10 steps of the oracle in fp32 (the
    reference's arithmetic) and in fp64, plus every step from a common state (what separates free-running trajectories is
    Adam's eps on zero-gradient elements, not accuracy: see test_trajectories_of_the_three_arithmetics).
"""
import numpy as np
import pytest
import torch

from conftest import DEFAULT_ARITH, load_golden
from oracle import ctn_oracle as O

pytestmark = pytest.mark.gpu

import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

DEV = "cuda:0"


def g(seed):
    return torch.Generator().manual_seed(seed)


def pad(t, Kp):
    out = t.new_zeros(t.shape[:-1] + (Kp,))
    out[..., : t.shape[-1]] = t
    return out


def dot_err(got, ref, scale):
    return float(((got.double().cpu() - ref).abs() / scale.clamp_min(1e-300)).max())


def amax_f(slots):
    """[M, 64] int32 bit patterns -> [M] float maxima"""
    return slots.view(torch.float32).amax(1)


SCENARIOS = {
    # name: (activation magnitude, gradient magnitude, weight magnitude, heavy tails, per-utterance factors)
    "unit": (1.0, 1.0, 0.05, False, None),
    "training": (3.0, 1e-7, 0.05, False, None),
    "tiny": (1.0, 1e-20, 1e-3, False, None),
    "large": (1e6, 1e5, 30.0, False, None),
    "heavy": (1.0, 1e-6, 0.05, True, None),
    "per_m": (1.0, 1e-3, 0.05, False, (-40, 12, 40)),
}


@pytest.mark.parametrize("scenario", sorted(SCENARIOS))
@pytest.mark.parametrize("M,B,H,K", [(3, 256, 512, 515), (3, 72, 132, 257)])
def test_h3_gemm_forms_against_fp64(scenario, M, B, H, K):
    sx, sg, sw, heavy, per_m = SCENARIOS[scenario]
    Kp = ops.padded_frames(K)
    xB, xH = torch.randn(M, B, K, generator=g(1)) * sx, torch.randn(M, H, K, generator=g(2)) * sx
    gB, gH = torch.randn(M, B, K, generator=g(3)) * sg, torch.randn(M, H, K, generator=g(4)) * sg
    if heavy:
        for i, t in enumerate((xB, xH, gB, gH)):
            t.mul_(torch.where(torch.rand(t.shape, generator=g(10 + i)) < 1e-4, 1e4, 1.0))
    if per_m is not None:
        f = torch.tensor([2.0 ** e for e in per_m]).view(M, 1, 1)
        xB, xH, gB, gH = xB * f, xH * f, gB * f.flip(0), gH * f.flip(0)
    xB, xH, gB, gH = (pad(t, Kp).to(DEV) for t in (xB, xH, gB, gH))
    w1 = (torch.randn(H, B, generator=g(5)) * sw).to(DEV)
    w2 = (torch.randn(B, H, generator=g(6)) * sw).to(DEV)
    a = torch.full((1,), 0.25, device=DEV)
    gam, bet = torch.randn(1, H, 1, generator=g(7)).to(DEV), torch.randn(1, H, 1, generator=g(8)).to(DEV)
    p1, p2 = ops.h3_pieces(w1, H, B, False), ops.h3_pieces(w2, B, H, False)
    q2, q1 = ops.h3_pieces(w2, H, B, True), ops.h3_pieces(w1, B, H, True)
    axB, axH, agB, agH = (ops.absmax_rows(t) for t in (xB, xH, gB, gH))
    gbm = ops.absmax_of(gam, bet)
    assert torch.equal(gbm.cpu(), torch.stack([gam.abs().max(), bet.abs().max()]).cpu())
    for t, am in ((xB, axB), (gH, agH)):
        assert torch.equal(amax_f(am).cpu(), t.abs().amax((1, 2)).cpu())                   # exact
    d = lambda t: t.double().cpu()  # noqa: E731

    def check(got, fp32_fn, ref, scale):
        e = dot_err(got, ref, scale)
        if e >= 6e-7:                       # only where the fp32 MFMA itself is above the limit on this data
            with ctn.gemm_arithmetic("fp32"):
                e32 = dot_err(fp32_fn(), ref, scale)
            assert e <= 1.25 * e32, (e, e32)

    # K1: h1 = W1 x, statistics of prelu(h1)
    ref = torch.einsum("rc,mck->mrk", d(w1), d(xB))
    sc = torch.einsum("rc,mck->mrk", d(w1).abs(), d(xB).abs())
    out, part = ops.pw_gemm_h3(p1, xB, H, B, K, axB, epi_alpha=a)
    check(out, lambda: ops.pw_gemm(w1, xB, H, B, K)[0], ref, sc)
    assert float(out[..., K:].abs().max()) == 0.0
    pre = torch.where(ref >= 0, ref, 0.25 * ref)[..., :K]
    s = part.sum(1).cpu()
    assert torch.allclose(s[:, 0], pre.sum((1, 2)), rtol=1e-5, atol=1e-5 * float(pre.abs().sum((1, 2)).max()))
    assert torch.allclose(s[:, 1], (pre ** 2).sum((1, 2)), rtol=1e-5)
    # B5: dx = W1^T dh1 + dy, maximum of the result tracked by the epilogue
    ref5 = torch.einsum("cr,mck->mrk", d(w1), d(gH)) + d(gB)
    sc5 = torch.einsum("cr,mck->mrk", d(w1).abs(), d(gH).abs()) + d(gB).abs()
    oam = torch.zeros(M, ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
    out5, _ = ops.pw_gemm_h3(q1, gH, B, H, K, agH, residual=gB, out_amax=oam)
    check(out5, lambda: ops.pw_gemm(w1, gH, B, H, K, trans_w=True, residual=gB)[0], ref5, sc5)
    assert torch.equal(amax_f(oam), out5.abs().amax((1, 2)))                               # exact: the maximum of what was stored
    # B1: dn2 = W2^T dy with the gLN-backward sums
    prx = torch.where(xH >= 0, xH, 0.25 * xH).double()
    cnt = H * K
    mean = prx[..., :K].sum((1, 2)) / cnt
    var = (prx[..., :K] ** 2).sum((1, 2)) / cnt - mean * mean
    rstd = 1.0 / torch.sqrt(var + 1e-8)
    ms = torch.stack([mean, rstd], -1).float().contiguous()
    refb1 = torch.einsum("cr,mck->mrk", d(w2), d(gB))
    scb1 = torch.einsum("cr,mck->mrk", d(w2).abs(), d(gB).abs())
    dn2, bpart = ops.pw_dgrad_gln_h3(q2, gB, H, B, K, xH, gam, a, ms, agB)
    check(dn2, lambda: ops.pw_dgrad_gln(w2, gB, H, B, K, xH, gam, a, ms)[0], refb1, scb1)
    xhat = ((prx - mean[:, None, None]) * rstd[:, None, None]).cpu()
    gd = d(gam) * refb1
    S = bpart.sum(1).cpu()
    for got, want, mag in ((S[:, 0], gd[..., :K].sum((1, 2)), gd[..., :K].abs().sum((1, 2))),
                           (S[:, 1], (gd * xhat)[..., :K].sum((1, 2)), (gd * xhat)[..., :K].abs().sum((1, 2)))):
        assert float(((got - want).abs() / mag.clamp_min(1e-300)).max()) < 1e-5
    # K3: out = W2 gLN(prelu(d)) + x
    st2 = torch.stack([prx[..., :K].sum((1, 2)), (prx[..., :K] ** 2).sum((1, 2))], -1).reshape(M, 1, 2).contiguous()
    nrm = d(gam) * xhat + d(bet)
    nrm[..., K:] = 0
    ref3 = torch.einsum("rc,mck->mrk", d(w2), nrm) + d(xB)
    sc3 = torch.einsum("rc,mck->mrk", d(w2).abs(), nrm.abs()) + d(xB).abs()
    ms_out = torch.empty(M, 2, device=DEV)
    out3, _ = ops.pw_gemm_h3(p2, xH, B, H, K, axH, pro=(st2, gam, bet, a), gbmax=gbm, residual=xB, ms_out=ms_out)
    check(out3, lambda: ops.pw_gemm(w2, xH, B, H, K, pro=(st2, gam, bet, a), residual=xB)[0], ref3, sc3)
    assert torch.allclose(ms_out.cpu(), ms.cpu(), rtol=2e-6)
    # weight gradients
    refw = torch.einsum("mrk,mck->rc", d(gH), d(xB))
    scw = torch.einsum("mrk,mck->rc", d(gH).abs(), d(xB).abs())
    check(ops.pw_wgrad_h3(gH, xB, H, B, K, agH, axB), lambda: ops.pw_wgrad(gH, xB, H, B, K), refw, scw)
    refw2 = torch.einsum("mrk,mck->rc", d(gB), nrm)
    scw2 = torch.einsum("mrk,mck->rc", d(gB).abs(), nrm.abs())
    check(ops.pw_wgrad_h3(gB, xH, B, H, K, agB, axH, pro=(gam, bet, a, ms), gbmax=gbm),
          lambda: ops.pw_wgrad(gB, xH, B, H, K, pro=(gam, bet, a, ms)), refw2, scw2)


def test_h3_adversarial_coherent_low_pieces():
    """The worst case of the h3 product, built on purpose: every operand is POSITIVE and sits just below the midpoint between two
    fp16 grid points of its scaled binade, x = c (1 + j 2^-10 + 2^-11 - 2^-17), so that its low piece is +(2^-11 - 2^-17) c for every
    element -- the dropped products a1.b1 are then all positive and add up coherently: the result is LOW by
    2^-22 / ((1 + j_a 2^-10)(1 + j_b 2^-10)) of every product, 1.1e-7 of sum |a||b| on average over j (up to 2.4e-7 for operands at
    the bottom of their binade), where random data cancels the dropped terms.  The second pattern also puts the remainder r at its
    largest magnitude 2^-23 |x| (a tie of the second rounding, resolved the same way for every element).
    Asserted: the coherent term is there and has the predicted sign and size (mean signed error), and the maximum error -- the
    coherent term plus the accumulator roundings, which on all-positive data are relative to the whole running sum -- stays under
    the 6e-7 gate of every other GEMM test or within 1.25x of the fp32 MFMA's on the same data.
    Per product the bound is  |a.b - h3(a, b)| <= (4 + 2 + 2) 2^-24 |a.b| = 4.8e-7 |a.b|  (a1.b1 <= 2^-22, r_a.b and a.r_b <= 2^-23
    each; rms 1.2 2^-24 on random data): csrc/ctn_gemm_b3.h."""
    M, B, H, K = 2, 256, 512, 515
    Kp = ops.padded_frames(K)
    for tail, coherent in ((2.0 ** -11 - 2.0 ** -17, True), (2.0 ** -11 - 2.0 ** -22 - 2.0 ** -23, False)):
        jx = torch.randint(0, 1024, (M, B, K), generator=g(1)).double()
        jw = torch.randint(0, 1024, (H, B), generator=g(2)).double()
        x64 = 8.0 * (1.0 + jx * 2.0 ** -10 + tail)
        w64 = 2.0 ** -6 * (1.0 + jw * 2.0 ** -10 + tail)
        x, w = pad(x64.float(), Kp).to(DEV), w64.float().to(DEV)
        assert torch.equal(x[..., :K].double().cpu(), x64) and torch.equal(w.double().cpu(), w64)       # exactly representable in fp32
        ref = torch.einsum("rc,mck->mrk", w64, x64)                                  # all terms positive: sum |a||b| = the result
        out, _ = ops.pw_gemm_h3(ops.h3_pieces(w, H, B, False), x, H, B, K, ops.absmax_rows(x))
        rel = (out[..., :K].double().cpu() - ref) / ref
        err, signed = float(rel.abs().max()), float(rel.mean())
        with ctn.gemm_arithmetic("fp32"):
            e32 = float(((ops.pw_gemm(w, x, H, B, K)[0][..., :K].double().cpu() - ref) / ref).abs().max())
        print("h3 adversarial pattern (low piece = %.6e of the element): max error %.3e (fp32 MFMA %.3e), mean signed error %+.3e of sum |a||b|"
              % (tail, err, e32, signed))
        if coherent:
            assert -1.7e-7 < signed < -0.6e-7, signed           # predicted -2^-22 E[1 / (1 + u)]^2 = -1.1e-7
        assert abs(signed) < 4.8e-7
        assert err < 6e-7 or err <= 1.25 * e32, (tail, err, e32)
        # the weight gradient sees the same operands on both sides (frames as the contraction)
        gH = pad((2.0 ** -6 * (1.0 + torch.randint(0, 1024, (M, H, K), generator=g(3)).double() * 2.0 ** -10 + tail)).float(), Kp).to(DEV)
        refw = torch.einsum("mrk,mck->rc", gH[..., :K].double().cpu(), x64)
        dw = ops.pw_wgrad_h3(gH, x, H, B, K, ops.absmax_rows(gH), ops.absmax_rows(x))
        errw = float(((dw.double().cpu() - refw).abs() / refw).max())
        with ctn.gemm_arithmetic("fp32"):
            ew32 = float(((ops.pw_wgrad(gH, x, H, B, K).double().cpu() - refw).abs() / refw).max())
        assert errw < 6e-7 or errw <= 1.25 * ew32, (tail, errw, ew32)


@pytest.mark.parametrize("blocks", [512, 100])
def test_wave_specialised_kernel_equals_the_one_role_kernel(blocks):
    """pw_gemm_ws_kernel (csrc/ctn_gemm_ws.h; opt-in: ctn_tune("b3_ws", 1)) against the product kernel on the four forward /
    input-gradient forms at the paper shapes with M = 8 -- several tiles per persistent workgroup, a partial last round, two
    workgroups per CU (512) or one (100): outputs bitwise equal (same MFMA order per accumulator), tracked maxima equal,
    statistics partials equal to rounding (they are summed in another fixed order).  This is the case that exposed the
    store-data hazard of 16-byte buffer stores with a register soffset (csrc/ctn_gemm_common.h)."""
    M, B, H, K = 8, 256, 512, 3199
    Kp = ops.padded_frames(K)
    xB, xH = (pad(torch.randn(M, c, K, generator=g(i)), Kp).to(DEV) for i, c in ((1, B), (2, H)))
    w1, w2 = (torch.randn(H, B, generator=g(5)) * 0.05).to(DEV), (torch.randn(B, H, generator=g(6)) * 0.05).to(DEV)
    a = torch.full((1,), 0.25, device=DEV)
    gam, bet = torch.randn(1, H, 1, generator=g(7)).to(DEV), torch.randn(1, H, 1, generator=g(8)).to(DEV)
    ms = torch.tensor([[0.1, 1.3]] * M, device=DEV)
    pre = torch.where(xH >= 0, xH, 0.25 * xH).double()
    st2 = torch.stack([pre[..., :K].sum((1, 2)), (pre[..., :K] ** 2).sum((1, 2))], -1).reshape(M, 1, 2).contiguous()
    p1, p2 = ops.h3_pieces(w1, H, B, False), ops.h3_pieces(w2, B, H, False)
    q2, q1 = ops.h3_pieces(w2, H, B, True), ops.h3_pieces(w1, B, H, True)
    axB, axH, gbm = ops.absmax_rows(xB), ops.absmax_rows(xH), ops.absmax_of(gam, bet)

    def forms():
        oam3 = torch.zeros(M, ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
        oam5 = torch.zeros(M, ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
        ms_out = torch.empty(M, 2, device=DEV)
        k1, k1p = ops.pw_gemm_h3(p1, xB, H, B, K, axB, epi_alpha=a)
        k3, _ = ops.pw_gemm_h3(p2, xH, B, H, K, axH, pro=(st2, gam, bet, a), gbmax=gbm, residual=xB, ms_out=ms_out, out_amax=oam3)
        b1, b1p = ops.pw_dgrad_gln_h3(q2, xB, H, B, K, xH, gam, a, ms, axB)
        b5, _ = ops.pw_gemm_h3(q1, xH, B, H, K, axH, residual=xB, out_amax=oam5)
        torch.cuda.synchronize()
        return (k1, k3, b1, b5), (k1p.sum(1), b1p.sum(1), ms_out), (amax_f(oam3), amax_f(oam5))

    try:
        ctn.lib.call("ctn_tune", b"b3_ws", 0)
        ref = forms()
        ctn.lib.call("ctn_tune", b"b3_ws", 1)
        ctn.lib.call("ctn_tune", b"b3_ws_blocks", blocks)
        for _ in range(3):                  # (the hazard was intermittent)
            got = forms()
            for name, x, y in zip(("K1", "K3", "B1", "B5"), ref[0], got[0]):
                assert torch.equal(x, y), "%s: %d outputs differ" % (name, int((x != y).sum()))
            for x, y in zip(ref[1], got[1]):
                assert torch.allclose(x, y, rtol=1e-6, atol=1e-6 * float(x.abs().max()))
            for x, y in zip(ref[2], got[2]):
                assert torch.equal(x, y)
    finally:
        ctn.lib.call("ctn_tune", b"b3_ws", 0)
        ctn.lib.call("ctn_tune", b"b3_ws_blocks", 512)


def test_h3_producers_track_exact_maxima():
    """ctn_dw_fwd (statistics epilogue) and ctn_gln_prelu_bwd write the maximum of what they store, per utterance."""
    M, H, K = 3, 132, 700
    Kp = ops.padded_frames(K)
    y = pad(torch.randn(M, H, K, generator=g(1)) * torch.tensor([1e-6, 1.0, 3e4]).view(M, 1, 1), Kp).to(DEV)
    D = torch.randn(H, 1, 3, generator=g(2)).to(DEV)
    a = torch.full((1,), 0.25, device=DEV)
    gam, bet = torch.randn(1, H, 1, generator=g(3)).to(DEV), torch.randn(1, H, 1, generator=g(4)).to(DEV)
    pre = torch.where(y >= 0, y, 0.25 * y).double()
    st = torch.stack([pre[..., :K].sum((1, 2)), (pre[..., :K] ** 2).sum((1, 2))], -1).reshape(M, 1, 2).contiguous()
    for dil in (1, 4, 64):
        Z = torch.empty_like(y)
        ep = torch.empty((M, H, 2), dtype=torch.float64, device=DEV)
        am = torch.zeros(M, ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
        ctn.lib.call("ctn_dw_fwd", y.data_ptr(), Z.data_ptr(), D.data_ptr(), M, H, K, Kp, 3, dil, 0, st.data_ptr(), 1, gam.data_ptr(),
                     bet.data_ptr(), a.data_ptr(), 0, a.data_ptr(), ep.data_ptr(), am.data_ptr(), ops._stream())
        assert torch.equal(amax_f(am), Z.abs().amax((1, 2)))
    dn = pad(torch.randn(M, H, K, generator=g(5)) * torch.tensor([1e-12, 1.0, 1e3]).view(M, 1, 1), Kp).to(DEV)
    ms = torch.tensor([[0.1, 1.3]] * M, device=DEV)
    s1p = torch.randn(M, H, 2, generator=g(6), dtype=torch.float64).to(DEV) * 1e-3
    dY, dap = torch.empty_like(dn), torch.empty(M * H, device=DEV)
    am = torch.zeros(M, ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
    ctn.lib.call("ctn_gln_prelu_bwd", dn.data_ptr(), y.data_ptr(), dY.data_ptr(), M, H, K, Kp, gam.data_ptr(), a.data_ptr(), ms.data_ptr(),
                 s1p.data_ptr(), H, dap.data_ptr(), am.data_ptr(), ops._stream())
    assert torch.equal(amax_f(am), dY.abs().amax((1, 2)))
    # channel-wise LayerNorm (the causal config's norm): forward output and input gradient, fast (v4) and fallback kernels
    for Ch, Kc in ((H, 700), (36, 203)):
        Kpc = ops.padded_frames(Kc)
        yc = pad(torch.randn(M, Ch, Kc, generator=g(7)) * torch.tensor([1e-6, 1.0, 3e4]).view(M, 1, 1), Kpc).to(DEV)
        gc, bc = torch.randn(Ch, generator=g(8)).to(DEV), torch.randn(Ch, generator=g(9)).to(DEV)
        out, mean, rstd = torch.empty_like(yc), torch.empty(M, Kpc, device=DEV), torch.empty(M, Kpc, device=DEV)
        am = torch.zeros(M, ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
        ctn.lib.call("ctn_cln_fwd", yc.data_ptr(), out.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, Ch, Kc, Kpc, gc.data_ptr(), bc.data_ptr(),
                     a.data_ptr(), am.data_ptr(), ops._stream())
        assert torch.equal(amax_f(am), out.abs().amax((1, 2)))
        dO = pad(torch.randn(M, Ch, Kc, generator=g(10)), Kpc).to(DEV)
        dYc = torch.empty_like(yc)
        pc = torch.empty(ctn.lib.ctn_cln_bwd_pc_floats(M, Ch, Kpc), device=DEV)
        dapc = torch.empty(ctn.lib.ctn_cln_bwd_blocks(M, Kpc), device=DEV)
        am = torch.zeros(M, ops.AMAX_SLOTS, dtype=torch.int32, device=DEV)
        ctn.lib.call("ctn_cln_bwd", dO.data_ptr(), yc.data_ptr(), dYc.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, Ch, Kc, Kpc, gc.data_ptr(),
                     a.data_ptr(), 0, 0, dapc.data_ptr(), pc.data_ptr(), am.data_ptr(), ops._stream())
        assert torch.equal(amax_f(am), dYc.abs().amax((1, 2)))


def test_h3_edge_cases():
    """An all-zero utterance (no maximum to scale by), NaN / inf confined to their own utterance, and elements far below their
    utterance's maximum: absolute error <= 2^-38 of max |x_m| times the row sum of |w| (the documented graceful loss)."""
    M, B, H, K = 4, 64, 128, 300
    Kp = ops.padded_frames(K)
    x = pad(torch.randn(M, B, K, generator=g(1)), Kp)
    x[1] = 0.0
    x[2, 3, 7] = float("nan")
    x[3, 5, 9] = float("inf")
    x = x.to(DEV)
    w = (torch.randn(H, B, generator=g(2)) * 0.1).to(DEV)
    out, _ = ops.pw_gemm_h3(ops.h3_pieces(w, H, B, False), x, H, B, K, ops.absmax_rows(x))
    ref = torch.einsum("rc,mck->mrk", w.double().cpu(), x.double().cpu())
    sc = torch.einsum("rc,mck->mrk", w.double().cpu().abs(), x.double().cpu().abs())
    assert dot_err(out[0], ref[0], sc[0]) < 6e-7
    assert float(out[1].abs().max()) == 0.0
    assert not torch.isfinite(out[2]).all() and not torch.isfinite(out[3]).all()          # poisoned utterances stay poisoned ...
    assert torch.isfinite(out[0]).all()                                                   # ... and alone
    # one huge element per utterance, everything else 2^-30 below it
    y = pad(torch.randn(M, B, K, generator=g(3)) * 2.0 ** -30, Kp)
    y[:, 0, 0] = 1.0
    y = y.to(DEV)
    outy, _ = ops.pw_gemm_h3(ops.h3_pieces(w, H, B, False), y, H, B, K, ops.absmax_rows(y))
    refy = torch.einsum("rc,mck->mrk", w.double().cpu(), y.double().cpu())
    bound = 2.0 ** -38 * w.double().cpu().abs().sum(1).view(1, H, 1) + 6e-7 * torch.einsum("rc,mck->mrk", w.double().cpu().abs(), y.double().cpu().abs())
    assert bool(((outy.double().cpu() - refy).abs() <= bound).all())
    # an all-zero weight matrix
    z = torch.zeros(H, B, device=DEV)
    outz, _ = ops.pw_gemm_h3(ops.h3_pieces(z, H, B, False), x[:1], H, B, K, ops.absmax_rows(x[:1]))
    assert float(outz.abs().max()) == 0.0


def _paper_step_setup(M):
    cfg = O.Config(N=256, L=20, B=256, H=512, P=3, X=8, R=4, C=2)
    torch.manual_seed(0)
    m = ctn.ConvTasNet(cfg.N, cfg.L, cfg.B, cfg.H, cfg.P, cfg.X, cfg.R, cfg.C).to(DEV)
    mix, lens, src = O.synth_batch(0, M, 32000)
    return cfg, m, mix, lens, src


def test_h3_paper_config_gradients_against_the_fp64_oracle():
    """One training step of BASELINE configs[1] (M = 1): every gradient against the CPU oracle run in fp64 on the same weights,
    under h3, b6 and the fp32 MFMA.  h3 must be at least as close to fp64 as the fp32 matrix instruction is."""
    cfg, m, mix, lens, src = _paper_step_setup(1)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    sd = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    loss_ref = O.cal_loss(src.double(), O.forward(cfg, sd, mix.double()), lens)[0]
    loss_ref.backward()
    tot = sum(float((v.grad ** 2).sum()) for v in sd.values() if v.grad is not None) ** 0.5
    err = {}
    for arith in ("h3", "b6", "fp32"):
        with ctn.gemm_arithmetic(arith):
            m.zero_grad()
            loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
            loss.backward()
            ops.join_side_stream()
            e2 = sum(float(((p.grad.double().cpu() - sd[k].grad) ** 2).sum()) for k, p in m.named_parameters())
            err[arith] = (e2 ** 0.5 / tot, abs(float(loss.detach()) - float(loss_ref.detach())))
    print("paper config, |g - g_fp64| / |g_fp64| and loss error [dB]: %s" % err)
    assert err["h3"][1] < 1e-3 and err["h3"][0] < 5e-5
    assert err["h3"][0] <= 1.1 * err["fp32"][0], err


def test_paper_config_gradients_at_the_bench_batch():
    """One training step of BASELINE configs[1] on the bench's batch (M = 8 x 4 s) under h3, b6 and the fp32 MFMA against the fp64
    oracle gradient of tests/golden/paper_grad_fp64_m8.npz (oracle/make_grad_golden.py, generated in the build container): the loss,
    every 997th gradient element and the norm of every tensor.  Observed on the MI355X over ALL elements
    (benchmarks/arith_grad_families.py, profiles/r04_b_grad_families.txt): |g - g64| / |g64| = 3.1e-6 (h3), 2.0e-5 (b6), 1.7e-5
    (fp32 MFMA) -- and 2.7e-6 / 1.8e-5 / 1.5e-5 at M = 1 and 2: the error does not grow with the batch under any arithmetic."""
    gd = load_golden("paper_grad_fp64_m8")
    stride = int(gd["stride"])
    cfg, m, mix, lens, src = _paper_step_setup(int(gd["M"]))
    g64 = gd["g"].astype(np.float64)
    err = {}
    for arith, bound in (("h3", 1e-5), ("b6", 6e-5), ("fp32", 5e-5)):
        with ctn.gemm_arithmetic(arith):
            m.zero_grad()
            loss = ctn.cal_loss(src.to(DEV), m(mix.to(DEV)), lens.to(DEV))[0]
            loss.backward()
            ops.join_side_stream()
            g = torch.cat([p.grad.detach().reshape(-1) for _, p in m.named_parameters()])
            e = float(np.linalg.norm(g[::stride].double().cpu().numpy() - g64) / np.linalg.norm(g64))
            norms = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
            en = float(np.max(np.abs(norms - gd["norms"]) / (gd["norms"] + 1e-6 * gd["norms"].max())))
            err[arith] = (e, abs(float(loss.detach()) - float(gd["loss"])), en)
            assert err[arith][1] < 1e-3 and e < bound and en < 5e-3, (arith, err[arith])      # observed: 2.9e-6 / 1.6e-5 / 1.3e-5; norms 1.3e-4 / 1.4e-3 / 1.6e-3
    print("paper config, M = 8: (|g - g_fp64| / |g_fp64| on every %dth element, loss error [dB], worst tensor-norm error) %s" % (stride, err))
    assert err["h3"][0] <= 1.1 * err["fp32"][0], err


def test_trajectories_of_the_three_arithmetics():
    """10 optimiser steps (fwd + PIT loss + bwd + clip(5) + Adam, lr 1e-3, eps 1e-8: src/solver.py:181-198) of the paper config on the
    bench's batch under the fp32 MFMA, b6 and h3, from the same weights on the same data, against
      (a) the CPU oracle in FP32 -- the reference's own arithmetic (tests/golden/paper_traj_cpu_fp32.npz, oracle/make_traj_fp32_golden.py),
      (b) the CPU oracle in fp64 (tests/golden/paper_traj_fp64.npz),
      (c) each other step by step FROM THE SAME STATE (the fp32-MFMA run's parameters and Adam moments before every step).
    What separates such runs (benchmarks/traj_adam_diag.py, traj_elem_diag.py): not the size of the gradient
    error (1e-5 .. 3e-4 relative at every step under every arithmetic) but Adam with eps = 1e-8 on elements whose gradient is
    (almost) zero.  A mask channel whose ReLU is off for every frame has an exactly zero gradient and zero Adam moments; the first
    step at which one frame switches it on gives its 256 weights a gradient of ~1e-5 of the typical size, and Adam turns that into a
    FULL-SIZE step lr * sign(g) -- 5.2e-4 per weight.  On this batch that happens at step 5 (row 187 of the mask convolution): its
    sign is decided by which frames are barely on, i.e. by differences of ~1e-7 in the stack's output; b6, the fp32 MFMA and the CPU
    in fp32 see -1.4e-5 there, h3 (and, by its losses, the fp64 run) +5.1e-5, the channel comes fully alive in one family of runs
    and dies again in the other, and the loss of step 6 differs by 3.9e-3 dB.  The same run-to-run sensitivity exists BETWEEN fp32
    implementations on other data; it is a property of the reference's optimiser rule, not an accuracy ranking.  Hence:
      (a) every arithmetic follows the reference arithmetic's trajectory within 4e-4 dB up to step 5 (observed <= 2.2e-4); from step 6
          on a run is within 2e-4 dB of the fixtures if it is on their side of the event and 3.9e-3 dB away otherwise: within 8e-3 dB
          is asserted.  Two runs of the REFERENCE'S OWN fp32 arithmetic land on different sides: the CPU oracle with 8 threads (this
          fixture) gives -0.310169 at step 6, with 16 threads on the GPU box's host -0.306235;
      (b) the same against the fp64 trajectory (the fp64 run is on the 8-thread CPU run's side);
      (c) from the same state, every step's loss agrees within 1e-4 dB (observed 4e-6) and the parameter update within 2 % (observed
          0.2 % for both b6 and h3: the sign noise of Adam on near-zero gradients) -- the bound that catches a drift of an
          arithmetic against the reference's without depending on which side of an event a free-running trajectory falls.
    Which GPU arithmetic is on which side is NOT a property of the arithmetic: it changed within round 4 when an unrelated kernel
    changed its summation order (comment at the assertion below; benchmarks/traj_repro.py shows each build's runs are bitwise
    reproducible run to run, across processes and with a NaN-dirtied allocator)."""
    from conv_tasnet_amd.optim import FlatAdam
    from conv_tasnet_amd.train import SyntheticLoader
    g64, g32 = load_golden("paper_traj_fp64"), load_golden("paper_traj_cpu_fp32")
    steps, stride = int(g64["steps"]), int(g64["stride"])
    assert int(g32["steps"]) == steps and int(g32["M"]) == int(g64["M"])
    mix, lens, src = next(iter(SyntheticLoader(1, int(g64["M"]), samples=32000)))
    mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)

    def one_step(m, opt):
        opt.zero_grad()
        loss = ctn.cal_loss(src, m(mix), lens)[0]
        loss.backward()
        opt.step(max_grad_norm=5.0)
        return float(loss.detach())

    dev, snaps = {}, []
    for arith in ("fp32", "b6", "h3"):
        ctn.set_gemm_arith(arith)
        torch.manual_seed(0)
        m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
        opt = FlatAdam(m.parameters(), lr=1e-3)
        losses = []
        for s in range(steps):
            if arith == "fp32":
                snaps.append([opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), s, None, None])
            losses.append(one_step(m, opt))
            if arith == "fp32":
                snaps[s][4], snaps[s][5] = losses[-1], opt.flat_params - snaps[s][0]
        p = torch.cat([q.detach().reshape(-1) for _, q in m.named_parameters()])[::stride].double().cpu().numpy()
        d32 = [abs(a - b) for a, b in zip(losses, g32["losses"])]
        dev[arith] = dict(fp64=max(abs(a - b) for a, b in zip(losses, g64["losses"])), early=max(d32[:6]), late=max(d32[6:]),
                          p64=float(np.linalg.norm(p - g64["p_final"].astype(np.float64)) / float(g64["travelled"])) * stride ** 0.5)
        if arith != "fp32":                 # (c): every step again from the fp32-MFMA run's state
            worst_l, worst_u = 0.0, 0.0
            for p0, m1, m2, s, l32, u32 in snaps:
                opt.flat_params.copy_(p0); opt.exp_avg.copy_(m1); opt.exp_avg_sq.copy_(m2); opt._step = s
                l = one_step(m, opt)
                worst_l = max(worst_l, abs(l - l32))
                worst_u = max(worst_u, float((opt.flat_params - p0 - u32).double().norm() / u32.double().norm()))
            dev[arith].update(same_state_loss=worst_l, same_state_update=worst_u)
    ctn.set_gemm_arith(DEFAULT_ARITH)
    print("10 steps of the paper config: %s" % dev)
    assert g64["losses"][-1] < g64["losses"][0] - 1.0
    for arith in ("fp32", "b6", "h3"):
        assert dev[arith]["early"] < 4e-4, (arith, dev[arith])
        # Behind the event a run is on one side or the other, and WHICH side changes with any last-bit change of any kernel: with
        # round 4's first build h3 was on the fixtures' side (1e-5 dB from fp64) and b6 / fp32 MFMA on the other; after the input
        # channel-wise LayerNorm went to 16-frame workgroups (another summation order of its per-frame sums) b6 is on the fixtures'
        # side (1.4e-4) and h3 / fp32 MFMA on the other (3.9e-3).  So only the loose bound is asserted for free-running runs.
        assert dev[arith]["late"] < 8e-3 and dev[arith]["fp64"] < 8e-3, (arith, dev[arith])
    for arith in ("b6", "h3"):
        assert dev[arith]["same_state_loss"] < 1e-4 and dev[arith]["same_state_update"] < 2e-2, (arith, dev[arith])
