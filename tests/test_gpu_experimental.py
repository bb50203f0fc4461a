"""Opt-in test of the split-bf16 experiment (include/ctn_hip_experimental.h).  Needs a library built with
CTN_BUILD_X6=1 and CTN_EXPERIMENTAL=1 in the environment; skipped otherwise (the default build does not ship it)."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("CTN_EXPERIMENTAL") != "1", reason="experimental build only")]

import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from test_gpu_parity import DEV, g, pad, rel_err  # noqa: E402


def test_split_bf16_presplit_gemm_p6():
    """Both operands pre-split into three bf16 planes (groundwork for the round-2 pipeline): fp32-level accuracy,
    per-utterance weights, row bias masked to k < K, fp32 + plane outputs that reconstruct exactly."""
    M, R, Cn, K = 2, 132, 72, 331
    Kp = ops.padded_frames(K)
    W = torch.randn(M, R, Cn, generator=g(1)) * 0.1
    X = pad(torch.randn(M, Cn, K, generator=g(2)), Kp)
    bias = torch.randn(M, R, generator=g(3))
    res = pad(torch.randn(M, R, K, generator=g(4)), Kp)
    Cnp = ctn.lib.ctn_split_cols(Cn)
    Wp = torch.empty((M, 3, R, Cnp), dtype=torch.bfloat16, device=DEV)
    Wd = W.to(DEV)
    for m in range(M):
        ctn.lib.call("ctn_split_bf16", Wd[m].data_ptr(), Wp[m].data_ptr(), R, Cn, 0, 0)
    Xd = X.to(DEV)
    Xp = torch.empty((3, M, Cn, Kp), dtype=torch.bfloat16, device=DEV)
    ctn.lib.call("ctn_split_act", Xd.data_ptr(), Xp.data_ptr(), Xd.numel(), 0)
    assert torch.equal(Xp.float().sum(0), Xd)                       # the 3-way split is exact
    out = torch.empty((M, R, Kp), device=DEV)
    outp = torch.empty((3, M, R, Kp), dtype=torch.bfloat16, device=DEV)
    ctn.lib.call("ctn_pw_gemm_p6", Wp.data_ptr(), 1, Xp.data_ptr(), out.data_ptr(), outp.data_ptr(), M, R, Cn, K, Kp,
                 bias.to(DEV).data_ptr(), res.to(DEV).data_ptr(), 0, 0, 0, 0, 0, 0, 0, 0, 0)
    ref = torch.einsum("moi,mik->mok", W.double(), X.double()) + res.double()
    ref[..., :K] += bias.double().unsqueeze(-1)
    assert rel_err(out, ref) < 1e-6
    assert torch.equal((outp[0].float() + outp[1].float()) + outp[2].float(), out)
    assert float(out[..., K:].abs().max()) == 0.0
