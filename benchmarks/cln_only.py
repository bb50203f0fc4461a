#!/usr/bin/env python
"""Time the channel-wise LayerNorm kernels (PReLU fused) at the causal paper shape [8, 512, 3199]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

M, H, K = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 3199
Kp = ops.padded_frames(K)
dev = "cuda:0"
if len(sys.argv) > 2:                   # frames per workgroup of the v4 kernels (ctn_tune "cln_fr": 16 | 32)
    import conv_tasnet_amd as ctn
    ctn.lib.call("ctn_tune", b"cln_fr", int(sys.argv[2]))
    if len(sys.argv) > 3:               # 0: the general backward kernel instead of the specialised one
        ctn.lib.call("ctn_tune", b"cln_lean", int(sys.argv[3]))
y = torch.randn(M, H, Kp, device=dev); y[..., K:] = 0
dout = torch.randn(M, H, Kp, device=dev); dout[..., K:] = 0
g, b = torch.randn(1, H, 1, device=dev), torch.randn(1, H, 1, device=dev)
a = torch.full((1,), 0.25, device=dev)


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
    return e0.elapsed_time(e1) / iters * 1e3


out, mean, rstd = ops.cln_fwd(y, g, b, a, K)
nbytes = M * H * Kp * 4
t = timeit(lambda: ops.cln_fwd(y, g, b, a, K))
print("cln_fwd  Ch=%d: %6.1f us  %.2f TB/s (1 read + 1 write)" % (H, t, 2 * nbytes / t / 1e6))
t = timeit(lambda: ops.cln_bwd(dout, y, mean, rstd, g, a, K))
print("cln_bwd  Ch=%d: %6.1f us  %.2f TB/s (params: 2 reads; dx: 2 reads + 1 write; + reductions)" % (H, t, 5 * nbytes / t / 1e6))
