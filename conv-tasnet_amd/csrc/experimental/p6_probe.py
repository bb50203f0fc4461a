#!/usr/bin/env python
"""Does the split-bf16 GEMM pay once BOTH operands are pre-split in HBM?  Accuracy vs fp64 and time per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from gemm_sweep import timeit, M, K, Kp  # noqa: E402

dev = "cuda:0"
lib = ctn.lib


def planes_of(x):
    p = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=dev)
    lib.call("ctn_split_act", x.data_ptr(), p.data_ptr(), x.numel(), 0)
    return p


for (R, Cn) in ((512, 256), (256, 512)):
    W = torch.randn(R, Cn, device=dev) * 0.05
    X = torch.randn(M, Cn, Kp, device=dev)
    X[..., K:] = 0
    Wp = ops._split_planes(W, R, Cn, False)
    Xp = planes_of(X)
    res = torch.randn(M, R, Kp, device=dev)
    out = torch.empty(M, R, Kp, device=dev)
    outp = torch.empty(3, M, R, Kp, dtype=torch.bfloat16, device=dev)
    flop = 2.0 * R * Cn * M * K

    def run(planes_out=False, residual=False):
        lib.call("ctn_pw_gemm_p6", Wp.data_ptr(), 0, Xp.data_ptr(), out.data_ptr(), outp.data_ptr() if planes_out else 0,
                 M, R, Cn, K, Kp, 0, res.data_ptr() if residual else 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    run(True)
    torch.cuda.synchronize()
    ref = torch.einsum("oi,ik->ok", W.double().cpu(), X[0].double().cpu())
    err = float((out[0].double().cpu() - ref).abs().max() / ref.abs().max())
    rec = outp.float().sum(0)
    err_p = float((rec - out).abs().max() / out.abs().max())
    print("R%d C%d: rel err vs fp64 %.2e; planes reconstruct err %.1e" % (R, Cn, err, err_p), flush=True)
    for tid in (3, 1, 2, 0):
        lib.ctn_tune_pw_tile(tid)
        t0 = timeit(lambda: run())
        t1 = timeit(lambda: run(True, True))
        print("   tile %d: plain %6.1f us (%6.1f TF)   +residual +planes-out %6.1f us" % (tid, t0, flop / t0 / 1e6, t1), flush=True)
    lib.ctn_tune_pw_tile(-1)
    t = timeit(lambda: planes_of(X))
    print("   stand-alone activation split: %.1f us" % t)
