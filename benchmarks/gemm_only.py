#!/usr/bin/env python
"""One GEMM form of the composite stacks launched N times at the paper shapes (for rocprofv3 / PMC passes), under the library's
arithmetic (CTN_GEMM_ARITH; h3 = the ctn_*_h3 entry points).  usage: gemm_only.py K1|K3|B1|B5|W1|W2 [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conv_tasnet_amd.ops import _p, _b3_pieces  # noqa: E402

form = sys.argv[1] if len(sys.argv) > 1 else "K1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda:0"
M, K = 8, 3199
Kp = ops.padded_frames(K)
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
pre = torch.where(xH >= 0, xH, 0.25 * xH).double()
st2 = torch.stack([pre[..., :K].sum((1, 2)), (pre[..., :K] ** 2).sum((1, 2))], -1).reshape(M, 1, 2).contiguous()
outH = torch.empty(M, H, Kp, device=dev)
outB = torch.empty(M, B, Kp, device=dev)
p1, p2 = _b3_pieces(w1, H, B, False), _b3_pieces(w2, B, H, False)
q2, q1 = _b3_pieces(w2, H, B, True), _b3_pieces(w1, B, H, True)
part = torch.empty((M, ctn.lib.ctn_pw_stats_parts(M, H, Kp), 2), dtype=torch.float64, device=dev)
sm = ops._stream()
fns = {
    "K1": lambda: ctn.lib.call("ctn_pw_gemm", _p(p1), _p(xB), _p(outH), M, H, B, K, Kp, 2, None, 0, None, None, None, None, None, _p(a), _p(part), 0, sm),
    "K3": lambda: ctn.lib.call("ctn_pw_gemm", _p(p2), _p(xH), _p(outB), M, B, H, K, Kp, 2, _p(st2), 1, _p(g), _p(b), _p(a), None, _p(xB), None, None, 0, sm),
    "B1": lambda: ctn.lib.call("ctn_pw_dgrad_gln_planes", _p(q2), _p(xB), _p(outH), M, H, B, K, Kp, _p(xH), _p(g), _p(a), _p(ms), _p(part), sm),
    "B5": lambda: ctn.lib.call("ctn_pw_gemm", _p(q1), _p(xH), _p(outB), M, B, H, K, Kp, 2, None, 0, None, None, None, None, _p(xB), None, None, 0, sm),
    "W1": lambda: ops.pw_wgrad(xH, xB, H, B, K),
    "W2": lambda: ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms)),
}
if ctn.gemm_arith() == "h3":
    P1, P2 = ops.h3_pieces(w1, H, B, False), ops.h3_pieces(w2, B, H, False)
    Q2, Q1 = ops.h3_pieces(w2, H, B, True), ops.h3_pieces(w1, B, H, True)
    axB, axH, gbm = ops.absmax_rows(xB), ops.absmax_rows(xH), ops.absmax_of(g, b)
    oam = torch.zeros(M, ops.AMAX_SLOTS, dtype=torch.int32, device=dev)
    fns = {
        "K1": lambda: ctn.lib.call("ctn_pw_gemm_h3", _p(P1), _p(xB), _p(outH), M, H, B, K, Kp, None, 0, None, None, None, None, None, _p(a), _p(part), _p(axB), None, None, sm),
        "K3": lambda: ctn.lib.call("ctn_pw_gemm_h3", _p(P2), _p(xH), _p(outB), M, B, H, K, Kp, _p(st2), 1, _p(g), _p(b), _p(a), None, _p(xB), None, None, _p(axH), _p(gbm), _p(oam), sm),
        "B1": lambda: ctn.lib.call("ctn_pw_dgrad_gln_h3", _p(Q2), _p(xB), _p(outH), M, H, B, K, Kp, _p(xH), _p(g), _p(a), _p(ms), _p(part), _p(axB), sm),
        "B5": lambda: ctn.lib.call("ctn_pw_gemm_h3", _p(Q1), _p(xH), _p(outB), M, B, H, K, Kp, None, 0, None, None, None, None, _p(xB), None, None, _p(axH), None, _p(oam), sm),
        "W1": lambda: ops.pw_wgrad_h3(xH, xB, H, B, K, axH, axB),
        "W2": lambda: ops.pw_wgrad_h3(xB, xH, B, H, K, axB, axH, pro=(g, b, a, ms), gbmax=gbm),
    }
fn = fns[form]
for _ in range(n):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    fn()
e1.record()
torch.cuda.synchronize()
print("%s %.1f us" % (form, e0.elapsed_time(e1) / n * 1e3))
