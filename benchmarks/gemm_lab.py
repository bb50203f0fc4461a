#!/usr/bin/env python
"""Time the six GEMM forms of a TemporalBlock at the paper shapes, in isolation, for ONE build of the library
(CTN_LIB_PATH selects an experiment build, e.g. benchmarks/lab/libctn_skipepi.so).  Prints one line per kernel form:
us per launch, TFLOP/s, fraction of the 157.3 TF fp32-MFMA peak.  Usage: python benchmarks/gemm_lab.py [tag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

dev = "cuda:0"
M, K = 8, 3199
Kp = ops.padded_frames(K)
tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(ctn.LIB_PATH)


def timeit(fn, iters=30):
    for _ in range(5):
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
    torch.manual_seed(0)
    xB = torch.randn(M, B, Kp, device=dev); xB[..., K:] = 0
    xH = torch.randn(M, H, Kp, device=dev); xH[..., K:] = 0
    w1 = torch.randn(H, B, device=dev) * 0.05
    w2 = torch.randn(B, H, device=dev) * 0.05
    a = torch.full((1,), 0.25, device=dev)
    g = torch.randn(1, H, 1, device=dev)
    b = torch.randn(1, H, 1, device=dev)
    ms = torch.tensor([[0.1, 1.3]] * M, device=dev)
    flop = 2.0 * H * B * M * K
    _, st = ops.pw_gemm(w1, xB, H, B, K, epi_alpha=a)
    _, st2 = ops.dw_fwd(xH, torch.randn(H, 1, 3, device=dev), K, 1, False, epi_alpha=a)
    cases = {
        "K1 fwd1 stats   R512 C256": lambda: ops.pw_gemm(w1, xB, H, B, K, epi_alpha=a),
        "K3 fwd2 pro+res R256 C512": lambda: ops.pw_gemm(w2, xH, B, H, K, pro=(st2, g, b, a), residual=xB),
        "B1 dgrad2 gln   R512 C256": lambda: ops.pw_dgrad_gln(w2, xB, H, B, K, xH, g, a, ms),
        "B5 dgrad1 T+res R256 C512": lambda: ops.pw_gemm(w1, xH, B, H, K, trans_w=True, residual=xB),
        "B6 wgrad1 plain R512 C256": lambda: ops.pw_wgrad(xH, xB, H, B, K),
        "B2 wgrad2 pro   R256 C512": lambda: ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms)),
        "-- fwd1 plain   R512 C256": lambda: ops.pw_gemm(w1, xB, H, B, K),
    }
    if os.environ.get("CTN_PW_KERNEL", "1") == "2":       # persistent kernels: forward forms on a transposed weight copy
        w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
        cases.update({
            "K1t fwd1 stats  W^T      ": lambda: ops.pw_gemm(w1t, xB, H, B, K, trans_w=True, epi_alpha=a),
            "K3t fwd2 pro+res W^T     ": lambda: ops.pw_gemm(w2t, xH, B, H, K, trans_w=True, pro=(st2, g, b, a), residual=xB),
            "--t fwd1 plain  W^T      ": lambda: ops.pw_gemm(w1t, xB, H, B, K, trans_w=True),
        })
    import ctypes
    dll = ctn.lib.load()
    has_clk = hasattr(dll, "ctn_debug_read")           # -DCTN_EXP_CLOCK builds
    only = [c for c in os.environ.get("LAB_CASES", "").split(",") if c]
    for name, fn in cases.items():
        if only and not any(name.startswith(c) for c in only):
            continue
        us = timeit(fn)
        clk = ""
        if has_clk:
            buf = (ctypes.c_ulonglong * 2)()
            dll.ctn_debug_read(buf, 2)
            if buf[1]:
                clk = "  in-kernel clock %.2f GHz (wg 0 alive %.1f us)" % (buf[0] / buf[1] * 0.1, buf[1] / 100.0)
        print("%-22s %-28s %8.1f us %7.1f TFLOP/s  %.3f of peak%s" % (tag, name, us, flop / us / 1e6, flop / us / 1e6 / 157.3, clk), flush=True)


if __name__ == "__main__":
    main()
