#!/usr/bin/env python
"""What does range tracking cost its producers?  The same launches with and without an amax output (paper shapes, alone)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conv_tasnet_amd.ops import _p  # noqa: E402

dev = "cuda:0"
M, K = 8, 3199
Kp = ops.padded_frames(K)
B, H = 256, 512
torch.manual_seed(0)
xB = torch.randn(M, B, Kp, device=dev); xB[..., K:] = 0
xH = torch.randn(M, H, Kp, device=dev); xH[..., K:] = 0
yH = torch.randn(M, H, Kp, device=dev); yH[..., K:] = 0
oH = torch.empty_like(xH)
w1 = torch.randn(H, B, device=dev) * 0.05
a = torch.full((1,), 0.25, device=dev)
g = torch.randn(H, device=dev)
b = torch.randn(H, device=dev)
D = torch.randn(H, 3, device=dev)
ms = torch.tensor([[0.1, 1.3]] * M, device=dev)
am = torch.zeros(M, 64, dtype=torch.int32, device=dev)
sm = ops._stream()


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


pre = torch.where(xH >= 0, xH, 0.25 * xH).double()
st = torch.stack([pre[..., :K].sum((1, 2)), (pre[..., :K] ** 2).sum((1, 2))], -1).reshape(M, 1, 2).contiguous()
ep = torch.empty((M, H, 2), dtype=torch.float64, device=dev)
s1p = torch.randn(M, H, 2, dtype=torch.float64, device=dev)
dap = torch.empty(M * H, device=dev)
for dil in (1, 16, 128):
    for tag, amp in (("no amax", 0), ("amax", _p(am))):
        t = timeit(lambda: ctn.lib.call("ctn_dw_fwd", _p(xH), _p(oH), _p(D), M, H, K, Kp, 3, dil, 0, _p(st), 1, _p(g), _p(b), _p(a), None, _p(a), _p(ep), amp, sm))
        print("dw_fwd dil %3d %-8s %6.1f us" % (dil, tag, t), flush=True)
for tag, amp in (("no amax", 0), ("amax", _p(am))):
    t = timeit(lambda: ctn.lib.call("ctn_gln_prelu_bwd", _p(xH), _p(yH), _p(oH), M, H, K, Kp, _p(g), _p(a), _p(ms), _p(s1p), H, _p(dap), amp, sm))
    print("gln_prelu_bwd   %-8s %6.1f us" % (tag, t), flush=True)
q1 = ops.h3_pieces(w1, B, H, True)
axH = ops.absmax_rows(xH)
oB = torch.empty_like(xB)
for tag, amp in (("no amax", 0), ("amax", _p(am))):
    t = timeit(lambda: ctn.lib.call("ctn_pw_gemm_h3", _p(q1), _p(xH), _p(oB), M, B, H, K, Kp, None, 0, None, None, None, None, _p(xB), None, None, _p(axH), None, amp, sm))
    print("B5 h3 (residual) %-8s %6.1f us" % (tag, t), flush=True)
t = timeit(lambda: ops.absmax_rows(xB, out=am))
print("absmax_rows [8,256,3200] %6.1f us" % t)
t = timeit(lambda: ops.h3_pieces(w1, H, B, False))
print("h3_pieces one matrix (absmax + split + alloc) %6.1f us" % t)
