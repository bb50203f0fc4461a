"""CPU oracle for the Conv-TasNet hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product path (``conv-tasnet_amd/``) never does:
it calls the HIP library through the C ABI in ``include/ctn_hip.h`` and raises
when that library is missing.

What this is: a clean-room, *functional* restatement (stock torch CPU ops over
a flat ``state_dict``; no nn.Module tree) of the arithmetic the reference runs
for one training / inference step.  Each function cites the reference lines
it restates (paths relative to /root/reference).

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference
modules in the build container, runs them on seeded inputs and commits the
inputs/outputs/gradients under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those vectors.
The reference ships no golden vectors or asserting tests of its own
(SURVEY.md section 4), so those generated fixtures are the only pin.

All functions are dtype-generic (fp32 for parity, fp64 for error budgets).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-8  # src/conv_tasnet.py:10, src/pit_criterion.py:9


@dataclass(frozen=True)
class Config:
    """Hyper-parameters, named as ConvTasNet.__init__ does (src/conv_tasnet.py:14-33)."""
    N: int
    L: int
    B: int
    H: int
    P: int
    X: int
    R: int
    C: int
    norm_type: str = "gLN"
    causal: bool = False
    mask_nonlinear: str = "relu"

    @property
    def stride(self) -> int:
        return self.L // 2  # src/conv_tasnet.py:106

    def frames(self, T: int) -> int:
        return (T - self.L) // self.stride + 1  # conv arithmetic, SURVEY App. B


# --------------------------------------------------------------------------
# parameter schema (measured from the reference state_dict)
# --------------------------------------------------------------------------
def block_keys(cfg: Config, r: int, x: int) -> Dict[str, str]:
    """state_dict keys of TemporalBlock (r, x).  src/conv_tasnet.py:218-278."""
    p = f"separator.network.2.{r}.{x}.net."
    bn = cfg.norm_type == "BN"         # nn.BatchNorm1d names its affine pair weight / bias (:309)
    gname, bname = ("weight", "bias") if bn else ("gamma", "beta")
    k = {"w1": p + "0.weight", "a1": p + "1.weight",
         "g1": p + "2." + gname, "b1": p + "2." + bname,
         "dw": p + "3.net.0.weight"}
    o = 1 if cfg.causal else 0  # Chomp1d shifts the integer names (:264-269)
    k["a2"] = p + f"3.net.{1 + o}.weight"
    k["g2"] = p + f"3.net.{2 + o}." + gname
    k["b2"] = p + f"3.net.{2 + o}." + bname
    k["w2"] = p + f"3.net.{3 + o}.weight"
    if bn:
        k["n1"] = p + "2."                  # + running_mean / running_var / num_batches_tracked
        k["n2"] = p + f"3.net.{2 + o}."
    return k


def param_shapes(cfg: Config) -> "Dict[str, Tuple[int, ...]]":
    """Ordered name -> shape, in nn.Module.parameters() order of the reference."""
    s: Dict[str, Tuple[int, ...]] = {}
    s["encoder.conv1d_U.weight"] = (cfg.N, 1, cfg.L)
    s["separator.network.0.gamma"] = (1, cfg.N, 1)
    s["separator.network.0.beta"] = (1, cfg.N, 1)
    s["separator.network.1.weight"] = (cfg.B, cfg.N, 1)
    for r in range(cfg.R):
        for x in range(cfg.X):
            k = block_keys(cfg, r, x)
            gshape = (cfg.H,) if cfg.norm_type == "BN" else (1, cfg.H, 1)
            s[k["w1"]] = (cfg.H, cfg.B, 1)
            s[k["a1"]] = (1,)
            s[k["g1"]] = gshape
            s[k["b1"]] = gshape
            s[k["dw"]] = (cfg.H, 1, cfg.P)
            s[k["a2"]] = (1,)
            s[k["g2"]] = gshape
            s[k["b2"]] = gshape
            s[k["w2"]] = (cfg.B, cfg.H, 1)
    s["separator.network.3.weight"] = (cfg.C * cfg.N, cfg.B, 1)
    s["decoder.basis_signals.weight"] = (cfg.L, cfg.N)
    return s


def init_buffers(cfg: Config, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """nn.BatchNorm1d's buffers at construction (norm_type="BN" only): running_mean 0, running_var 1, count 0."""
    out: Dict[str, torch.Tensor] = {}
    if cfg.norm_type != "BN":
        return out
    for r in range(cfg.R):
        for x in range(cfg.X):
            k = block_keys(cfg, r, x)
            for pre in (k["n1"], k["n2"]):
                out[pre + "running_mean"] = torch.zeros(cfg.H, dtype=dtype)
                out[pre + "running_var"] = torch.ones(cfg.H, dtype=dtype)
                out[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return out


def init_params(cfg: Config, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Reference-compatible initial values, *not* bit-identical RNG streams.

    Restates the init rule of src/conv_tasnet.py:41-43: every tensor with
    dim()>1 -- including the [1,Ch,1] gamma/beta of each norm --
    is xavier-normal; PReLU slopes stay 0.25 (nn.PReLU default, :224,:259).
    """
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg).items():
        if len(shape) > 1:
            rf = 1
            for d in shape[2:]:
                rf *= d
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
            std = math.sqrt(2.0 / (fan_in + fan_out))
            out[name] = (torch.randn(shape, generator=g, dtype=torch.float64) * std).to(dtype)
        elif shape == (1,):
            out[name] = torch.full(shape, 0.25, dtype=dtype)
        else:                       # BatchNorm1d affine pair: weight 1, bias 0 (dim 1: untouched by the xavier loop)
            out[name] = torch.ones(shape, dtype=dtype) if name.endswith("weight") else torch.zeros(shape, dtype=dtype)
    out.update(init_buffers(cfg, dtype))
    return out


# --------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------
def encoder(mix: torch.Tensor, U: torch.Tensor, stride: int) -> torch.Tensor:
    """w[m,n,k] = relu(sum_l U[n,0,l] x[m,k*S+l]).  src/conv_tasnet.py:106,119-120."""
    return F.conv1d(mix.unsqueeze(1), U, stride=stride).clamp_min(0)


def cln(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """Per-frame LN over channels, biased variance.  src/conv_tasnet.py:332-334.

    NOT cumulative."""
    mu = y.mean(dim=1, keepdim=True)
    var = ((y - mu) ** 2).mean(dim=1, keepdim=True)
    return gamma * (y - mu) / torch.sqrt(var + EPS) + beta


def gln(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """Per-utterance LN over (channels, frames), two-pass.  src/conv_tasnet.py:358-360."""
    mu = y.mean(dim=(1, 2), keepdim=True)
    var = ((y - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    return gamma * (y - mu) / torch.sqrt(var + EPS) + beta


BN_EPS, BN_MOMENTUM = 1e-5, 0.1     # nn.BatchNorm1d defaults, which src/conv_tasnet.py:309 takes


def bn(y: torch.Tensor, weight, bias, sd, prefix: str, training: bool) -> torch.Tensor:
    """nn.BatchNorm1d on [M, Ch, K]: statistics per channel over (M, K).  src/conv_tasnet.py:305-309.

    training: batch mean / biased variance normalise; the running pair in ``sd`` is blended in place with the
    batch mean / UNBIASED variance and the batch counter advances.  eval: the running pair normalises."""
    rm, rv = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    if training:
        n = y.shape[0] * y.shape[2]
        mu = y.mean(dim=(0, 2))
        var = ((y - mu.view(1, -1, 1)) ** 2).mean(dim=(0, 2))
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mu.detach().to(rm.dtype))
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * (var.detach() * n / max(n - 1, 1)).to(rv.dtype))
            sd[prefix + "num_batches_tracked"] += 1
    else:
        mu, var = rm.to(y.dtype), rv.to(y.dtype)
    return (y - mu.view(1, -1, 1)) / torch.sqrt(var.view(1, -1, 1) + BN_EPS) * weight.view(1, -1, 1) + bias.view(1, -1, 1)


def norm(cfg: Config, y, gamma, beta, sd=None, prefix=None, training=True):
    """chose_norm, src/conv_tasnet.py:298-310."""
    if cfg.norm_type == "gLN":
        return gln(y, gamma, beta)
    if cfg.norm_type == "cLN":
        return cln(y, gamma, beta)
    return bn(y, gamma, beta, sd, prefix, training)


def prelu(y: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """Single-slope PReLU.  src/conv_tasnet.py:224,259."""
    return torch.where(y >= 0, y, a * y)


def pointwise(y: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """1x1 conv == per-frame GEMM, no bias.  src/conv_tasnet.py:174,191,223,262."""
    return torch.einsum("oi,mik->mok", W[:, :, 0], y)


def depthwise(y: torch.Tensor, D: torch.Tensor, dilation: int, causal: bool) -> torch.Tensor:
    """z[m,h,k] = sum_j D[h,0,j] y[m,h,k + j*d - pad_left], zeros outside [0,K).

    Non-causal: pad (P-1)d/2 both sides; causal: pad (P-1)d both sides then drop
    the last (P-1)d frames == left pad only.  src/conv_tasnet.py:182,253-256,295.
    """
    P = D.shape[-1]
    total = (P - 1) * dilation
    left = total if causal else total // 2
    right = 0 if causal else total - left
    yp = F.pad(y, (left, right))
    K = y.shape[-1]
    z = torch.zeros_like(y)
    for j in range(P):
        z = z + D[:, 0, j].view(1, -1, 1) * yp[:, :, j * dilation: j * dilation + K]
    return z


def temporal_block(cfg: Config, x: torch.Tensor, sd, r: int, xi: int, training: bool = True) -> torch.Tensor:
    """x + pw2(norm(prelu(dw(norm(prelu(pw1(x)))))))  src/conv_tasnet.py:231-243,265-269."""
    k = block_keys(cfg, r, xi)
    h = pointwise(x, sd[k["w1"]])
    h = norm(cfg, prelu(h, sd[k["a1"]]), sd[k["g1"]], sd[k["b1"]], sd, k.get("n1"), training)
    h = depthwise(h, sd[k["dw"]], 2 ** xi, cfg.causal)
    h = norm(cfg, prelu(h, sd[k["a2"]]), sd[k["g2"]], sd[k["b2"]], sd, k.get("n2"), training)
    return x + pointwise(h, sd[k["w2"]])


def separator(cfg: Config, w: torch.Tensor, sd, training: bool = True) -> torch.Tensor:
    """mixture_w [M,N,K] -> est_mask [M,C,N,K].  src/conv_tasnet.py:198-215.

    The first norm is always channel-wise (:172)."""
    y = cln(w, sd["separator.network.0.gamma"], sd["separator.network.0.beta"])
    y = pointwise(y, sd["separator.network.1.weight"])
    for r in range(cfg.R):
        for xi in range(cfg.X):
            y = temporal_block(cfg, y, sd, r, xi, training)
    score = pointwise(y, sd["separator.network.3.weight"])
    M, _, K = score.shape
    score = score.reshape(M, cfg.C, cfg.N, K)
    if cfg.mask_nonlinear == "relu":
        return score.clamp_min(0)
    if cfg.mask_nonlinear == "softmax":
        return torch.softmax(score, dim=1)
    raise ValueError("Unsupported mask non-linear function")  # :214


def overlap_and_add(frames: torch.Tensor, step: int) -> torch.Tensor:
    """out[..., j*step + l] += frames[..., j, l].  src/utils.py:9-47.

    Deterministic shifted-slab formulation instead of the reference's
    gcd-subframe index_add_."""
    *outer, K, L = frames.shape
    q = -(-L // step)
    fp = F.pad(frames, (0, q * step - L)).reshape(*outer, K, q, step)
    out = frames.new_zeros(*outer, K + q - 1, step)
    for i in range(q):
        out[..., i:i + K, :] += fp[..., :, i, :]
    return out.reshape(*outer, -1)[..., : (K - 1) * step + L]


def decoder(cfg: Config, w: torch.Tensor, mask: torch.Tensor, V: torch.Tensor) -> torch.Tensor:
    """est[m,c,:] = OLA_k( V @ (w[m,:,k] * mask[m,c,:,k]) ).  src/conv_tasnet.py:140-145."""
    sw = w.unsqueeze(1) * mask                      # [M,C,N,K]
    fr = torch.einsum("ln,mcnk->mckl", V, sw)       # [M,C,K,L]
    return overlap_and_add(fr, cfg.stride)


def forward(cfg: Config, sd, mixture: torch.Tensor, training: bool = True) -> torch.Tensor:
    """mixture [M,T] -> est_source [M,C,T], right-padded with zeros.  src/conv_tasnet.py:45-60.

    ``training`` only matters for norm_type="BN" (batch vs running statistics)."""
    w = encoder(mixture, sd["encoder.conv1d_U.weight"], cfg.stride)
    mask = separator(cfg, w, sd, training)
    est = decoder(cfg, w, mask, sd["decoder.basis_signals.weight"])
    return F.pad(est, (0, mixture.shape[-1] - est.shape[-1]))


# --------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------
def length_mask(lengths: torch.Tensor, T: int, dtype) -> torch.Tensor:
    """[B,1,T] 1 where t < len[b].  src/pit_criterion.py:102-114."""
    t = torch.arange(T).view(1, 1, T)
    return (t < lengths.view(-1, 1, 1)).to(dtype)


def pairwise_si_snr(source, est, lengths):
    """snr[b,i,j]: estimate i against target j, dB.  src/pit_criterion.py:36-63."""
    Bn, C, T = source.shape
    mk = length_mask(lengths, T, source.dtype)
    est = est * mk                                   # :38 (in place there)
    n = lengths.view(-1, 1, 1).to(source.dtype)
    s0 = (source - source.sum(2, keepdim=True) / n) * mk    # :41-47
    e0 = (est - est.sum(2, keepdim=True) / n) * mk          # :42-48
    dot = torch.einsum("bit,bjt->bij", e0, s0)              # :56
    en = (s0 ** 2).sum(2) + EPS                             # :57  [B,C]
    proj = dot.unsqueeze(-1) * s0.unsqueeze(1) / en.view(Bn, 1, C, 1)   # :58
    noise = e0.unsqueeze(2) - proj                           # :60
    ratio = (proj ** 2).sum(3) / ((noise ** 2).sum(3) + EPS)  # :62
    return 10 * torch.log10(ratio + EPS), est                # :63


def permutations(C: int) -> torch.Tensor:
    """[C!,C] in itertools order; perms[p][i]=j pairs estimate i with target j (:67)."""
    return torch.tensor(list(itertools.permutations(range(C))), dtype=torch.long)


def si_snr_pit(source, est, lengths):
    """(max_snr [B,1], perms [C!,C], idx [B], masked est).  src/pit_criterion.py:27-77."""
    snr, est_m = pairwise_si_snr(source, est, lengths)
    C = source.shape[1]
    perms = permutations(C)
    rows = torch.arange(C)
    score = torch.stack([snr[:, rows, p].sum(1) for p in perms], dim=1)   # :72 einsum
    best, idx = score.max(dim=1, keepdim=True)   # first max wins ties, like argmax (:73)
    return best / C, perms, idx.view(-1), est_m


def reorder(est, perms, idx):
    """out[b,c] = est[b, perm_b[c]] -- applies the perm, not its inverse, exactly
    as src/pit_criterion.py:80-99 does (SURVEY a13 flags this for 3-cycles)."""
    sel = perms[idx]                                  # [B,C]
    return torch.gather(est, 1, sel.unsqueeze(-1).expand_as(est))


def cal_loss(source, est, lengths):
    """src/pit_criterion.py:12-24.  Functional: returns the masked estimate
    instead of mutating the argument."""
    max_snr, perms, idx, est_m = si_snr_pit(source, est, lengths)
    loss = 0 - max_snr.mean()
    return loss, max_snr, est_m, reorder(est_m, perms, idx)


def cal_sisnr_np(ref_sig: np.ndarray, out_sig: np.ndarray, eps: float = 1e-8) -> float:
    """numpy SI-SNR of the evaluator.  src/evaluate.py:114-130."""
    r = ref_sig - ref_sig.mean()
    o = out_sig - out_sig.mean()
    proj = (r * o).sum() * r / ((r ** 2).sum() + eps)
    noise = o - proj
    return float(10 * np.log((proj ** 2).sum() / ((noise ** 2).sum() + eps) + eps) / np.log(10.0))


def cal_sisnri_np(src_ref, src_est, mix) -> float:
    """2-speaker SI-SNR improvement.  src/evaluate.py:94-111."""
    a = [cal_sisnr_np(src_ref[c], src_est[c]) - cal_sisnr_np(src_ref[c], mix) for c in range(2)]
    return (a[0] + a[1]) / 2


# --------------------------------------------------------------------------
# one optimiser step (src/solver.py:188-196 with Adam from src/train.py:92-95)
# --------------------------------------------------------------------------
def clip_coef(grads: List[torch.Tensor], max_norm: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch.nn.utils.clip_grad_norm_ rule: total L2, coef = min(1, max/(total+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).to(grads[0].dtype)
    return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def adam_update(p, g, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) single-tensor arithmetic order."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def train_step(cfg: Config, sd, opt_state, mixture, source, lengths, lr=1e-3, max_norm=5.0):
    """fwd -> loss -> bwd -> clip -> Adam, in place on ``sd``.  Returns loss (python float)."""
    names = list(param_shapes(cfg).keys())
    leaves = {n: sd[n].detach().requires_grad_(True) for n in names}
    leaves.update({n: v for n, v in sd.items() if n not in leaves})      # BatchNorm buffers, updated in place
    est = forward(cfg, leaves, mixture)
    loss, _, _, _ = cal_loss(source, est, lengths)
    grads = torch.autograd.grad(loss, [leaves[n] for n in names])
    _, coef = clip_coef(list(grads), max_norm)
    opt_state["step"] = opt_state.get("step", 0) + 1
    with torch.no_grad():
        for n, g in zip(names, grads):
            st = opt_state.setdefault(n, {"m": torch.zeros_like(sd[n]), "v": torch.zeros_like(sd[n])})
            adam_update(sd[n], g * coef, st["m"], st["v"], opt_state["step"], lr)
    return float(loss)


# --------------------------------------------------------------------------
# deterministic synthetic workload
# --------------------------------------------------------------------------
def synth_batch(first_utt: int, M: int, T: int, C: int = 2, sr: int = 8000):
    """Harmonic 'speakers' + noise: (mixture [M,T], lengths [M], sources [M,C,T]), fp32."""
    t = torch.arange(T, dtype=torch.float64) / sr
    src = torch.empty(M, C, T, dtype=torch.float64)
    for i in range(M):
        g = torch.Generator().manual_seed(1234 + first_utt + i)
        for c in range(C):
            f0 = 80 + 320 * torch.rand(1, generator=g, dtype=torch.float64)
            ph = 2 * math.pi * torch.rand(3, generator=g, dtype=torch.float64)
            s = torch.zeros(T, dtype=torch.float64)
            for h, a in enumerate((1.0, 0.5, 0.25)):
                s = s + a * torch.sin(2 * math.pi * (h + 1) * f0 * t + ph[h])
            src[i, c] = s + 0.01 * torch.randn(T, generator=g, dtype=torch.float64)
    src = src.float()
    return src.sum(1), torch.full((M,), T, dtype=torch.long), src
