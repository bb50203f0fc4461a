"""CPU: pin the oracle restatement against vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from oracle import ctn_oracle as O
from conftest import load_golden


def _cfg(g):
    N, L, B, H, P, X, R, C = [int(v) for v in g["cfg"]]
    return O.Config(N, L, B, H, P, X, R, C, norm_type=str(g.get("norm_type", "gLN")),
                    causal=bool(int(g.get("causal", 0))), mask_nonlinear=str(g.get("mask_nonlinear", "relu")))


def _sd(g, prefix="p:"):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


MODEL_CASES = ["model_tiny_gln", "model_tiny_cln_causal", "model_c3_softmax", "model_c3_relu_x4"]


@pytest.mark.parametrize("name", MODEL_CASES)
def test_forward_loss_grads_match_reference(name):
    g = load_golden(name)
    cfg = _cfg(g)
    sd = _sd(g)
    assert list(sd.keys()) == list(O.param_shapes(cfg).keys())
    for k, shp in O.param_shapes(cfg).items():
        assert tuple(sd[k].shape) == shp, k
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    mix = torch.from_numpy(g["mixture"])
    src = torch.from_numpy(g["source"])
    lens = torch.from_numpy(g["lengths"])
    est = O.forward(cfg, leaves, mix)
    ref_est = torch.from_numpy(g["est_source_raw"])
    scale = ref_est.abs().max().item()
    assert (est - ref_est).abs().max().item() <= 2e-5 * scale
    loss, max_snr, est_m, reord = O.cal_loss(src, est, lens)
    # 1e-3 dB is the north-star budget; the restatement itself must sit far inside it
    assert abs(float(loss) - float(g["loss"])) < 2e-4
    np.testing.assert_allclose(max_snr.detach().numpy(), g["max_snr"], atol=2e-4)
    np.testing.assert_allclose(est_m.detach().numpy(), g["est_source_masked"], atol=2e-5 * scale)
    np.testing.assert_allclose(reord.detach().numpy(), g["reorder"], atol=2e-5 * scale)
    loss.backward()
    for k in sd:
        ref = g["g:" + k]
        got = leaves[k].grad.numpy()
        tol = 2e-3 * np.abs(ref).max() + 1e-7
        assert np.abs(got - ref).max() <= tol, (k, np.abs(got - ref).max(), np.abs(ref).max())


def test_intermediates_tiny():
    g = load_golden("model_tiny_gln")
    cfg = _cfg(g)
    sd = _sd(g)
    mix = torch.from_numpy(g["mixture"])
    w = O.encoder(mix, sd["encoder.conv1d_U.weight"], cfg.stride)
    np.testing.assert_allclose(w.numpy(), g["i_encoder"], atol=1e-5)
    y = O.cln(w, sd["separator.network.0.gamma"], sd["separator.network.0.beta"])
    np.testing.assert_allclose(y.numpy(), g["i_cln0"], atol=2e-5)
    y = O.pointwise(y, sd["separator.network.1.weight"])
    np.testing.assert_allclose(y.numpy(), g["i_bottleneck"], atol=2e-5)
    k = O.block_keys(cfg, 0, 0)
    h = O.pointwise(y, sd[k["w1"]])
    np.testing.assert_allclose(h.numpy(), g["i_block00_pw1"], atol=2e-5)
    n1 = O.gln(O.prelu(h, sd[k["a1"]]), sd[k["g1"]], sd[k["b1"]])
    np.testing.assert_allclose(n1.numpy(), g["i_block00_norm1"], atol=2e-5)
    d = O.depthwise(n1, sd[k["dw"]], 1, False)
    np.testing.assert_allclose(d.numpy(), g["i_block00_dw"], atol=2e-5)
    out = O.temporal_block(cfg, y, sd, 0, 0)
    np.testing.assert_allclose(out.numpy(), g["i_block00"], atol=5e-5)
    mask = O.separator(cfg, w, sd)
    np.testing.assert_allclose(mask.numpy(), g["i_mask"], atol=1e-4 * np.abs(g["i_mask"]).max())


def test_pit_known_answer_from_reference_main():
    """Inputs of src/pit_criterion.py:117-133 (seed 123); SURVEY section 4 records loss 45.9221."""
    g = load_golden("pit_main_int")
    src = torch.from_numpy(g["source"]).float()
    est = torch.from_numpy(g["estimate"]).float()
    loss, max_snr, _, _ = O.cal_loss(src, est, torch.from_numpy(g["lengths"]))
    assert abs(float(loss) - 45.9221) < 1e-3
    assert abs(float(loss) - float(g["loss"])) < 1e-3
    np.testing.assert_allclose(max_snr.numpy(), g["max_snr"], atol=1e-3)


@pytest.mark.parametrize("C", [2, 3])
def test_pit_float_ragged(C):
    g = load_golden("pit_float_c%d" % C)
    src = torch.from_numpy(g["source"])
    est = torch.from_numpy(g["estimate"]).requires_grad_(True)
    lens = torch.from_numpy(g["lengths"])
    loss, max_snr, est_m, reord = O.cal_loss(src, est, lens)
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    np.testing.assert_allclose(max_snr.detach().numpy(), g["max_snr"], atol=1e-4)
    _, perms, idx, _ = O.si_snr_pit(src, est.detach(), lens)
    assert np.array_equal(perms.numpy(), g["perms"])
    assert np.array_equal(idx.numpy(), g["idx"])
    np.testing.assert_allclose(est_m.detach().numpy(), g["est_masked"], atol=1e-6)
    np.testing.assert_allclose(reord.detach().numpy(), g["reorder"], atol=1e-6)
    loss.backward()
    np.testing.assert_allclose(est.grad.numpy(), g["grad_estimate"], atol=2e-3 * np.abs(g["grad_estimate"]).max())


@pytest.mark.parametrize("name", ["ola_main_int", "ola_f_37_20_10", "ola_f_50_16_8", "ola_f_11_21_10"])
def test_overlap_and_add(name):
    g = load_golden(name)
    sig = torch.from_numpy(g["signal"]).double()
    out = O.overlap_and_add(sig, int(g["step"]))
    np.testing.assert_allclose(out.numpy(), g["result"].astype(np.float64), atol=1e-5)


def test_sisnr_numpy_matches_pairwise_diagonal():
    """src/evaluate.py:114-130 formula vs the torch pairwise SI-SNR on full-length signals."""
    g = load_golden("pit_float_c2")
    src, est = g["source"][:1], g["estimate"][:1]
    snr, _ = O.pairwise_si_snr(torch.from_numpy(src), torch.from_numpy(est), torch.tensor([src.shape[-1]]))
    for c in range(2):
        assert abs(O.cal_sisnr_np(src[0, c].astype(np.float64), est[0, c].astype(np.float64)) - float(snr[0, c, c])) < 1e-3


def test_train_steps_follow_reference_solver():
    """Adam + clip(5) trajectory of the reference Solver (src/solver.py:181-198) on a 3-batch epoch."""
    g = load_golden("solver_traj")
    cfg = _cfg(g)
    sd = _sd(g, "p0:")
    T = int(g["T"])
    batches = [O.synth_batch(900 + 2 * i, 2, T) for i in range(3)]
    state = {}
    seen = []
    for _ in range(int(g["epochs"])):
        for mix, lens, src in batches:
            seen.append(O.train_step(cfg, sd, state, mix, src, lens))
        with torch.no_grad():
            mix, lens, src = batches[0]
            seen.append(float(O.cal_loss(src, O.forward(cfg, sd, mix), lens)[0]))
    np.testing.assert_allclose(np.array(seen), g["iter_losses"], atol=2e-3)
    for k, v in _sd(g, "p1:").items():
        np.testing.assert_allclose(sd[k].numpy(), v.numpy(), atol=2e-4, err_msg=k)
    # epoch average divides by (n+1)
    n = 3
    ep0 = sum(g["iter_losses"][:n]) / (n + 1)
    assert abs(ep0 - float(g["tr_loss"][0])) < 1e-4


@pytest.mark.parametrize("name", ["model_tiny_bn", "model_tiny_bn_causal"])
def test_batchnorm_variant_matches_reference(name):
    """norm_type="BN": training-mode forward / loss / gradients / running statistics and the eval-mode forward."""
    g = load_golden(name)
    cfg = _cfg(g)
    sd0 = _sd(g, "p0:")
    shapes = O.param_shapes(cfg)
    assert [k for k in sd0 if k in shapes] == list(shapes.keys())
    assert set(sd0) == set(shapes) | set(O.init_buffers(cfg))
    for k, v in O.init_buffers(cfg).items():
        assert torch.equal(sd0[k].to(v.dtype), v), k
    leaves = {k: (v.clone().requires_grad_(True) if k in shapes else v.clone()) for k, v in sd0.items()}
    mix, src, lens = (torch.from_numpy(g[k]) for k in ("mixture", "source", "lengths"))
    est = O.forward(cfg, leaves, mix, training=True)
    ref = torch.from_numpy(g["est_source_raw"])
    scale = ref.abs().max().item()
    assert (est - ref).abs().max().item() <= 2e-5 * scale
    loss, max_snr, _, _ = O.cal_loss(src, est, lens)
    assert abs(float(loss) - float(g["loss"])) < 2e-4
    loss.backward()
    for k in shapes:
        r = g["g:" + k]
        d = np.abs(leaves[k].grad.numpy() - r).max()
        assert d <= 2e-3 * np.abs(r).max() + 1e-7, (k, d)
    for k in O.init_buffers(cfg):       # running statistics after one batch
        np.testing.assert_allclose(leaves[k].numpy(), g["p1:" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    with torch.no_grad():
        ev = O.forward(cfg, leaves, mix, training=False)
    ref_ev = torch.from_numpy(g["est_source_eval"])
    assert (ev - ref_ev).abs().max().item() <= 2e-5 * ref_ev.abs().max().item()
