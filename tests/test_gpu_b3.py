"""GPU: the opt-in "b3" GEMM arithmetic (two bf16 pieces per fp32 operand, three bf16 MFMAs per product step: ~16 significant
bits per product -- NOT reference precision, never the default, never the benchmarked arithmetic).

It is kept as a documented speed / precision trade and gets its own, explicitly stated limits here, each
about 3x what was observed on the MI355X (profiles/r03_a_b6_check.txt, benchmarks/b3_grad_err.py):

  * every GEMM form against fp64: error <= 8e-6 of sum |a||b| (observed 2.6e-6; fp32 MFMA and the default b6: 4e-7);
  * golden models (reference-generated fixtures): loss within the north star's 1e-3 dB (observed <= 9e-5 dB), separated waveforms
    within 3.5e-5 of the largest sample (observed 1.1e-5), every gradient within 3e-3 of its largest element (observed 9e-4);
  * a TRAJECTORY: 10 optimiser steps of the paper config on the bench's batch of 8 under b3 against the same 10 steps under the
    default arithmetic -- per-step loss within 1e-3 dB, parameters after the 10th step within a stated fraction of the distance
    they travelled: 16-bit products do not drift a training run.
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


@pytest.fixture
def b3():
    ctn.set_gemm_arith("b3")
    yield
    ctn.set_gemm_arith(DEFAULT_ARITH)


def g(seed):
    return torch.Generator().manual_seed(seed)


def pad(t, Kp):
    out = t.new_zeros(t.shape[:-1] + (Kp,))
    out[..., : t.shape[-1]] = t
    return out


def dot_err(got, ref, scale):
    """max |got - ref| in units of sum |a||b| (the natural scale of a dot product's rounding error)."""
    return float(((got.double().cpu() - ref).abs() / scale.clamp_min(1e-30)).max())


@pytest.mark.parametrize("M,R,Cn,K", [(2, 256, 512, 515), (2, 512, 256, 1000), (3, 132, 72, 257)])
def test_b3_gemm_forms_against_fp64(b3, M, R, Cn, K):
    Kp = ops.padded_frames(K)
    W = torch.randn(R, Cn, generator=g(1)) * 0.1
    X = pad(torch.randn(M, Cn, K, generator=g(2)), Kp)
    dO = pad(torch.randn(M, R, K, generator=g(3)), Kp)
    ref = torch.einsum("oi,mik->mok", W.double(), X.double())
    sc = torch.einsum("oi,mik->mok", W.double().abs(), X.double().abs())
    for trans in (False, True):
        Wd = (W.t().contiguous() if trans else W).to(DEV)
        out, _ = ops.pw_gemm(Wd, X.to(DEV), R, Cn, K, trans_w=trans)          # pre-split weight pieces (R >= 64)
        assert dot_err(out[..., :K], ref[..., :K], sc[..., :K]) < 8e-6
    refw = torch.einsum("mrk,mck->rc", dO.double(), X.double())
    scw = torch.einsum("mrk,mck->rc", dO.double().abs(), X.double().abs())
    assert dot_err(ops.pw_wgrad(dO.to(DEV), X.to(DEV), R, Cn, K), refw, scw) < 8e-6
    with ctn.gemm_arithmetic(DEFAULT_ARITH):                                   # and the default arithmetic on the same data: fp32 level
        out6, _ = ops.pw_gemm(W.to(DEV), X.to(DEV), R, Cn, K)
        assert dot_err(out6[..., :K], ref[..., :K], sc[..., :K]) < 6e-7


def _load_model(gd):
    from test_gpu_parity import _load_model as lm
    return lm(gd)


@pytest.mark.parametrize("name", ["model_tiny_gln", "model_tiny_cln_causal", "model_c3_softmax", "model_c3_relu_x4"])
def test_b3_model_against_reference_golden(b3, name):
    gd = load_golden(name)
    m = _load_model(gd)
    mix, src, lens = (torch.from_numpy(gd[k]).to(DEV) for k in ("mixture", "source", "lengths"))
    est = m(mix)
    ref = torch.from_numpy(gd["est_source_raw"])
    assert float((est.detach().cpu().double() - ref.double()).abs().max() / ref.abs().max()) < 3.5e-5
    loss = ctn.cal_loss(src, est, lens)[0]
    assert abs(float(loss.detach()) - float(gd["loss"])) < 1e-3                # north-star budget, dB
    loss.backward()
    for k, p in m.named_parameters():
        r = torch.from_numpy(gd["g:" + k]).double()
        e = float((p.grad.detach().cpu().double() - r).abs().max() / (r.abs().max() + 1e-30))
        assert e < 3e-3, (k, e)


def test_b3_does_not_drift_a_paper_config_trajectory():
    """10 optimiser steps (fwd + PIT loss + bwd + clip(5) + Adam, lr 1e-3) of the paper config on the bench's batch under b3 and
    under b6 (three bf16 pieces: the arithmetic b3 is a truncation of), from the same weights on the same data."""
    from conv_tasnet_amd.optim import FlatAdam
    from conv_tasnet_amd.train import SyntheticLoader
    mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
    mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
    runs = {}
    for arith in ("b6", "b3"):
        ctn.set_gemm_arith(arith)
        torch.manual_seed(0)
        m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
        opt = FlatAdam(m.parameters(), lr=1e-3)
        p0 = opt.flat_params.detach().clone()
        losses = []
        for _ in range(10):
            opt.zero_grad()
            loss = ctn.cal_loss(src, m(mix), lens)[0]
            loss.backward()
            opt.step(max_grad_norm=5.0)
            losses.append(float(loss.detach()))
        runs[arith] = (losses, opt.flat_params.detach().clone(), p0)
    ctn.set_gemm_arith(DEFAULT_ARITH)
    (l6, p6, p0), (l3, p3, _) = runs["b6"], runs["b3"]
    dl = max(abs(a - b) for a, b in zip(l6, l3))
    travelled = float((p6 - p0).double().norm())
    apart = float((p3 - p6).double().norm())
    print("b3 vs b6 over 10 steps: max |loss difference| %.2e dB, |p_b3 - p| / |p - p0| = %.2e, losses %s" % (dl, apart / travelled, l6))
    assert l6[-1] < l6[0] - 1.0                                               # the run trains (the SI-SNR loss falls by > 1 dB)
    assert dl < 1e-3, dl
    assert apart < 5e-3 * travelled, (apart, travelled)            # observed 1.4e-3 (max loss difference 8.8e-5 dB)
