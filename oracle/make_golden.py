#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY.  Needs /root/reference (read-only mount); it never
runs on the GPU box -- only the small .npz files it writes travel there.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

Every array stored is data (seeded inputs, reference outputs, reference
gradients).  No reference source text is copied anywhere.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.modules.setdefault("visdom", types.ModuleType("visdom"))  # src/solver.py:7 imports it unconditionally

import warnings  # noqa: E402
warnings.filterwarnings("ignore")

from src.conv_tasnet import ConvTasNet  # noqa: E402
from src.pit_criterion import cal_loss, cal_si_snr_with_pit  # noqa: E402
from src.utils import overlap_and_add  # noqa: E402
from src.solver import Solver  # noqa: E402
import src.solver as ref_solver_mod  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.ctn_oracle import synth_batch  # noqa: E402  (inputs only)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def cfg_arr(**kw):
    order = ["N", "L", "B", "H", "P", "X", "R", "C"]
    return np.array([kw[k] for k in order], dtype=np.int64)


def model_case(name, T, M, lengths, seed, norm_type="gLN", causal=False, mask="relu",
               with_intermediates=False, **hp):
    torch.manual_seed(seed)
    model = ConvTasNet(hp["N"], hp["L"], hp["B"], hp["H"], hp["P"], hp["X"], hp["R"], hp["C"],
                       norm_type=norm_type, causal=causal, mask_nonlinear=mask)
    mix, _, src = synth_batch(100 * seed, M, T, C=hp["C"])
    lens = torch.tensor(lengths, dtype=torch.long)
    # zero the padding like the data loader's pad_list does (src/data.py:320-331)
    for b, n in enumerate(lengths):
        mix[b, n:] = 0
        src[b, :, n:] = 0
    arrs = {"cfg": cfg_arr(**hp), "norm_type": norm_type, "causal": int(causal), "mask_nonlinear": mask,
            "mixture": mix, "source": src, "lengths": lens}
    inter = {}
    if with_intermediates:
        def grab(tag):
            def hook(_m, _i, o):
                inter[tag] = o.detach().clone()
            return hook
        model.encoder.register_forward_hook(grab("i_encoder"))
        model.separator.network[0].register_forward_hook(grab("i_cln0"))
        model.separator.network[1].register_forward_hook(grab("i_bottleneck"))
        model.separator.network[2][0][0].register_forward_hook(grab("i_block00"))
        model.separator.network[2][0][0].net[0].register_forward_hook(grab("i_block00_pw1"))
        model.separator.network[2][0][0].net[2].register_forward_hook(grab("i_block00_norm1"))
        model.separator.network[2][0][0].net[3].net[0].register_forward_hook(grab("i_block00_dw"))
        model.separator.network[2].register_forward_hook(grab("i_tcn"))
        model.separator.register_forward_hook(grab("i_mask"))
    est = model(mix)
    arrs["est_source_raw"] = est.detach().clone()       # before cal_loss masks it in place
    loss, max_snr, est_masked, reorder = cal_loss(src, est, lens)
    loss.backward()
    arrs.update(loss=loss.detach(), max_snr=max_snr.detach(), est_source_masked=est_masked.detach(),
                reorder=reorder.detach())
    for k, v in model.state_dict().items():
        arrs["p:" + k] = v
    for k, p in model.named_parameters():
        arrs["g:" + k] = p.grad
    arrs.update(inter)
    save(name, **arrs)


def bn_case(name, T, M, seed, causal=False, **hp):
    """norm_type="BN" (chose_norm's else-branch, src/conv_tasnet.py:305-309): one training-mode forward/backward
    (batch statistics, running statistics updated) and one eval-mode forward (running statistics)."""
    torch.manual_seed(seed)
    model = ConvTasNet(hp["N"], hp["L"], hp["B"], hp["H"], hp["P"], hp["X"], hp["R"], hp["C"],
                       norm_type="BN", causal=causal, mask_nonlinear="relu")
    with torch.no_grad():       # move the affine parameters and slopes off their trivial defaults
        for k, p in model.named_parameters():
            if p.dim() == 1 and p.numel() > 1:
                p.add_(0.2 * torch.randn_like(p) if k.endswith("weight") else 0.1 * torch.randn_like(p))
    mix, lens, src = synth_batch(100 * seed, M, T, C=hp["C"])
    arrs = {"cfg": cfg_arr(**hp), "norm_type": "BN", "causal": int(causal), "mask_nonlinear": "relu",
            "mixture": mix, "source": src, "lengths": lens}
    for k, v in model.state_dict().items():
        arrs["p0:" + k] = v.clone()
    model.train()
    est = model(mix)
    arrs["est_source_raw"] = est.detach().clone()
    loss, max_snr, _, _ = cal_loss(src, est, lens)
    loss.backward()
    arrs.update(loss=loss.detach(), max_snr=max_snr.detach())
    for k, v in model.state_dict().items():
        arrs["p1:" + k] = v.clone()
    for k, p in model.named_parameters():
        arrs["g:" + k] = p.grad
    model.eval()
    with torch.no_grad():
        arrs["est_source_eval"] = model(mix).clone()
    save(name, **arrs)


def pit_cases():
    # (a) the reference's own __main__ smoke inputs (src/pit_criterion.py:117-133): seed 123, randint(4)
    torch.manual_seed(123)
    B, C, T = 2, 3, 32000
    source = torch.randint(4, (B, C, T))
    est = torch.randint(4, (B, C, T))
    source[1, :, -3:] = 0
    est[1, :, -3:] = 0
    lens = torch.LongTensor([T, T - 3])
    src_in, est_in = source.clone(), est.clone()
    loss, max_snr, est_m, reord = cal_loss(source, est, lens)
    save("pit_main_int", source=src_in.to(torch.int8), estimate=est_in.to(torch.int8), lengths=lens,
         loss=loss.float(), max_snr=max_snr.float())
    # (b) float, ragged, C=2 and C=3, with gradients wrt the estimate
    for C in (2, 3):
        torch.manual_seed(7 + C)
        B, T = 3, 4000
        src = torch.randn(B, C, T)
        perm = torch.randperm(C)
        est = (src[:, perm] + 0.3 * torch.randn(B, C, T)).requires_grad_(True)
        lens = torch.LongTensor([T, T - 123, T // 2])
        for b in range(B):
            src[b, :, lens[b]:] = 0
        est_in = est.detach().clone()
        e2 = est * 1.0  # non-leaf so the in-place mask is legal
        loss, max_snr, est_m, reord = cal_loss(src, e2, lens)
        loss.backward()
        ms, perms, idx = cal_si_snr_with_pit(src, est_in.clone(), lens)
        save("pit_float_c%d" % C, source=src, estimate=est_in, lengths=lens, loss=loss.detach(),
             max_snr=max_snr.detach(), est_masked=est_m.detach(), reorder=reord.detach(),
             perms=perms, idx=idx, grad_estimate=est.grad)


def ola_cases():
    torch.manual_seed(123)  # src/utils.py:70-77 __main__
    sig = torch.randint(5, (2, 2, 3, 4))
    save("ola_main_int", signal=sig, step=2, result=overlap_and_add(sig, 2))
    torch.manual_seed(5)
    for (K, L, s) in ((37, 20, 10), (50, 16, 8), (11, 21, 10)):
        sig = torch.randn(2, 3, K, L)
        save("ola_f_%d_%d_%d" % (K, L, s), signal=sig, step=s, result=overlap_and_add(sig, s))


def solver_case(name="solver_traj", hp=None, T=2005, epochs=4):
    """Reference Solver on CPU: DataParallel is a pass-through with 0 GPUs and supplies .module.
    solver_traj: the round-1 fixture (tiny widths: the product runs its fp32 kernels on every layer but one);
    solver_traj_wide: B = 64, H = 128 -- every 1x1 convolution of the stack has >= 64 rows, so the product's default h3
    arithmetic (and b6) is what the reference's recorded trajectory is compared with."""
    hp = dict(N=16, L=20, B=8, H=16, P=3, X=2, R=1, C=2) if hp is None else hp
    torch.manual_seed(3)
    model = ConvTasNet(hp["N"], hp["L"], hp["B"], hp["H"], hp["P"], hp["X"], hp["R"], hp["C"])
    init_sd = {k: v.clone() for k, v in model.state_dict().items()}
    wrapped = torch.nn.DataParallel(model)
    opt = torch.optim.Adam(wrapped.parameters(), lr=1e-3, weight_decay=0)
    batches = []
    for i in range(3):
        mix, lens, src = synth_batch(900 + 2 * i, 2, T)
        batches.append((mix, lens, src))
    cv = [(b[0].clone(), b[1].clone(), b[2].clone()) for b in batches[:1]]
    seen = []
    orig = ref_solver_mod.cal_loss

    def spy(*a, **k):
        out = orig(*a, **k)
        seen.append(float(out[0]))
        return out
    ref_solver_mod.cal_loss = spy
    import tempfile
    tmp = tempfile.mkdtemp()
    arg = (0, epochs, 1, 0, 5, tmp, 0, "", "final.pth.tar", 1000, 0, 0, "x")
    s = Solver({"tr_loader": batches, "cv_loader": cv}, wrapped, opt, arg)
    s.train()
    ref_solver_mod.cal_loss = orig
    arrs = {"cfg": cfg_arr(**hp), "T": T, "epochs": epochs, "iter_losses": np.array(seen, dtype=np.float64),
            "tr_loss": s.tr_loss.clone(), "cv_loss": s.cv_loss.clone(),
            "final_lr": opt.param_groups[0]["lr"]}
    for k, v in init_sd.items():
        arrs["p0:" + k] = v
    for k, v in model.state_dict().items():
        arrs["p1:" + k] = v
    pkg = torch.load(os.path.join(tmp, "final.pth.tar"), weights_only=False)
    arrs["pkg_keys"] = np.array(sorted(pkg.keys()))
    arrs["pkg_epoch"] = pkg["epoch"]
    save(name, **arrs)


def init_case():
    """Seeded constructor output: lets the build's constructor be checked bit-for-bit."""
    hp = dict(N=16, L=20, B=8, H=16, P=3, X=2, R=2, C=2)
    for causal, nt in ((False, "gLN"), (True, "cLN")):
        torch.manual_seed(11)
        m = ConvTasNet(hp["N"], hp["L"], hp["B"], hp["H"], hp["P"], hp["X"], hp["R"], hp["C"],
                       norm_type=nt, causal=causal)
        arrs = {"cfg": cfg_arr(**hp)}
        for k, v in m.state_dict().items():
            arrs["p:" + k] = v
        save("init_seed11_%s" % nt, **arrs)


def synth_wave(path, sr=8000):
    """Deterministic stand-in for a wav file: the file name encodes (id, speaker tag, n_samples)."""
    base = os.path.basename(path)[:-4]
    _, uid, tag, n = base.split("_")
    rs = np.random.RandomState(1000 * int(uid) + {"mix": 0, "s1": 1, "s2": 2}[tag])
    return rs.uniform(-1, 1, int(n)).astype(np.float32)


def data_case():
    """Minibatch planning + collation of the reference loaders (src/data.py) on a synthetic manifest.
    librosa is absent: it is stubbed (like visdom) with a loader that synthesises the waveform from the file name,
    so only the reference's own planning / segmenting / padding logic runs."""
    import json
    import tempfile
    fake = types.ModuleType("librosa")
    fake.load = lambda path, sr=None: (synth_wave(path), sr)
    sys.modules["librosa"] = fake
    from src.data import AudioDataset, _collate_fn
    sr, seg_s = 8000, 0.5
    lens = [21000, 13000, 12000, 9000, 8000, 8000, 7999, 6500, 4000, 4000, 3999, 3000, 12345, 4001, 16000, 2000, 5000]
    tmp = tempfile.mkdtemp()
    for tag in ("mix", "s1", "s2"):
        infos = [["/fake/utt_%d_%s_%d.wav" % (i, tag, n), n] for i, n in enumerate(lens)]
        with open(os.path.join(tmp, tag + ".json"), "w") as f:
            json.dump(infos, f)
    arrs = {"lens": np.array(lens), "sample_rate": sr, "segment": seg_s}
    for name, kw in (("tr_b3", dict(batch_size=3, segment=seg_s)), ("tr_b5", dict(batch_size=5, segment=seg_s)),
                     ("cv_b2", dict(batch_size=2, segment=-1, cv_maxlen=2.0)), ("cv_b4", dict(batch_size=4, segment=-1, cv_maxlen=8.0))):
        ds = AudioDataset(tmp, sample_rate=sr, **kw)
        plan_ids, plan_off, blens, bsum_mix, bsum_src = [], [0], [], [], []
        for mb in ds.minibatch:
            ids = [int(os.path.basename(info[0]).split("_")[1]) for info in mb[0]]
            plan_ids += ids
            plan_off.append(len(plan_ids))
            mix, ln, src = _collate_fn([mb])
            blens.append(np.pad(ln.numpy(), (0, 8 - len(ln)), constant_values=-1))
            bsum_mix.append(float(mix.double().sum()))
            bsum_src.append(float((src.double() * torch.arange(1, src.shape[1] + 1).view(1, -1, 1)).sum()))
            assert mix.shape[0] == src.shape[0] == len(ln) and src.shape[1] == 2
        arrs[name + ":ids"] = np.array(plan_ids)
        arrs[name + ":off"] = np.array(plan_off)
        arrs[name + ":lens"] = np.array(blens)
        arrs[name + ":sum_mix"] = np.array(bsum_mix)
        arrs[name + ":sum_src"] = np.array(bsum_src)
    save("data_plan", **arrs)


if __name__ == "__main__":
    if "--data-only" in sys.argv:
        data_case()
        sys.exit(0)
    tiny = dict(N=64, L=20, B=32, H=64, P=3, X=2, R=2, C=2)
    if "--solver-wide-only" in sys.argv:
        solver_case("solver_traj_wide", dict(N=64, L=20, B=64, H=128, P=3, X=2, R=2, C=2), T=4005, epochs=4)
        sys.exit(0)
    if "--bn-only" in sys.argv:
        bn_case("model_tiny_bn", T=4005, M=3, seed=5, **tiny)
        bn_case("model_tiny_bn_causal", T=3001, M=2, seed=6, causal=True, **tiny)
        sys.exit(0)
    model_case("model_tiny_gln", T=4005, M=2, lengths=[4005, 3777], seed=1, with_intermediates=True, **tiny)
    model_case("model_tiny_cln_causal", T=3001, M=2, lengths=[3001, 2500], seed=2, norm_type="cLN", causal=True,
               **tiny)
    model_case("model_c3_softmax", T=2400, M=2, lengths=[2400, 2400], seed=3, mask="softmax",
               N=32, L=16, B=16, H=32, P=3, X=3, R=1, C=3)
    model_case("model_c3_relu_x4", T=2400, M=1, lengths=[2211], seed=4,
               N=32, L=16, B=16, H=32, P=3, X=4, R=2, C=3)
    bn_case("model_tiny_bn", T=4005, M=3, seed=5, **tiny)
    bn_case("model_tiny_bn_causal", T=3001, M=2, seed=6, causal=True, **tiny)
    pit_cases()
    ola_cases()
    init_case()
    solver_case()
    solver_case("solver_traj_wide", dict(N=64, L=20, B=64, H=128, P=3, X=2, R=2, C=2), T=4005, epochs=4)
    data_case()
