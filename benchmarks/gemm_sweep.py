#!/usr/bin/env python
"""Sweep the 1x1-conv GEMM tile shapes at the paper-config shapes (MI355X).  Prints us and TFLOP/s per variant."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

dev = "cuda:0"
M, K = 8, 3199
Kp = ops.padded_frames(K)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    B, H = 256, 512
    xB = torch.randn(M, B, Kp, device=dev); xB[..., K:] = 0
    xH = torch.randn(M, H, Kp, device=dev); xH[..., K:] = 0
    w1 = torch.randn(H, B, device=dev) * 0.05
    w2 = torch.randn(B, H, device=dev) * 0.05
    a = torch.full((1,), 0.25, device=dev)
    g = torch.randn(1, H, 1, device=dev)
    b = torch.randn(1, H, 1, device=dev)
    ms = torch.tensor([[0.1, 1.3]] * M, device=dev)
    flop = 2.0 * H * B * M * K
    rows = []
    mode = ops.gemm_mode()
    # accuracy of this mode against fp64
    ref = torch.einsum("oi,mik->mok", w1.double().cpu(), xB[:1].double().cpu())
    got, _ = ops.pw_gemm(w1, xB[:1].contiguous(), H, B, K)
    err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
    print("mode %s: max rel err of W1.x vs fp64 = %.3e" % (mode, err), flush=True)
    for tid in ([int(t) for t in os.environ.get("SWEEP_TILES", "3,1,8,9,10").split(",")] if mode == "x6" else [int(t) for t in os.environ.get("SWEEP_TILES", "0,1,2,3,4,5,6,7,8,9").split(",")]):
        ctn.lib.ctn_tune_pw_tile(tid)
        _, st = ops.pw_gemm(w1, xB, H, B, K, epi_alpha=a)
        np2 = ctn.lib.ctn_pw_stats_parts(M, H, Kp)
        s2p = torch.empty((M, np2, 2), dtype=torch.float64, device=dev)
        dn = torch.empty((M, H, Kp), device=dev)
        cases = {
            "fwd1 plain   R512 C256": lambda: ops.pw_gemm(w1, xB, H, B, K),
            "fwd1 stats   R512 C256": lambda: ops.pw_gemm(w1, xB, H, B, K, epi_alpha=a),
            "fwd2 pro+res R256 C512": lambda: ops.pw_gemm(w2, xH, B, H, K, pro=(st, g, b, a), residual=xB),
            "dgrad2 gln   R512 C256": lambda: ops.pw_dgrad_gln(w2, xB, H, B, K, xH, g, a, ms),
            "dgrad1 T+res R256 C512": lambda: ops.pw_gemm(w1, xH, B, H, K, trans_w=True, residual=xB),
        }
        for name, fn in cases.items():
            us = timeit(fn)
            rows.append((tid, name, us, flop / us / 1e6))
    ctn.lib.ctn_tune_pw_tile(-1)
    for name, fn in {"wgrad plain  R512 C256": lambda: ops.pw_wgrad(xH, xB, H, B, K),
                     "wgrad pro    R256 C512": lambda: ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms))}.items():
        us = timeit(fn)
        rows.append((-1, name, us, flop / us / 1e6))
    names = ["128x128", "128x64", "64x128", "64x64", "128x64w", "64x64k32", "128x64k32", "128x128k32", "128x128w8", "128x64w8", "128x128ws", "-"]
    for tid, name, us, tf in rows:
        print("tile %-8s %-24s %8.1f us %7.1f TFLOP/s" % (names[tid], name, us, tf), flush=True)


if __name__ == "__main__":
    main()
