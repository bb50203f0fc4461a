#!/usr/bin/env python
"""Energy per launch of the split-bf16 GEMM experiments next to the fp32-MFMA product kernels (see power_lab.py).
The library is chosen by CTN_LIB_PATH (benchmarks/lab_x6.so: three pieces / six products, lab_x3.so: two pieces / three
products, both built with CTN_BUILD_X6=1); LABEL prefixes the case names."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conv_tasnet_amd.ops import _p  # noqa: E402

dev = "cuda:0"
M, K = 8, 3199
Kp = ops.padded_frames(K)
SECONDS = float(os.environ.get("SECONDS_PER_CASE", "2.5"))
LABEL = os.environ.get("LABEL", "x")
B, H = 256, 512
torch.manual_seed(0)
xB = torch.randn(M, B, Kp, device=dev); xB[..., K:] = 0
xH = torch.randn(M, H, Kp, device=dev); xH[..., K:] = 0
w1 = torch.randn(H, B, device=dev) * 0.05
w2 = torch.randn(B, H, device=dev) * 0.05
a = torch.full((1,), 0.25, device=dev)
g = torch.randn(1, H, 1, device=dev)
b = torch.randn(1, H, 1, device=dev)
D = torch.randn(H, 1, 3, device=dev)
ms = torch.tensor([[0.1, 1.3]] * M, device=dev)


def case(name, fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.time()
    while time.time() - t0 < SECONDS:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    t1 = time.time()
    print("case %-28s %.3f %.3f %d %.2f" % ((LABEL + "_" + name).replace(" ", "_"), t0, t1, n, (t1 - t0) / n * 1e6), flush=True)
    time.sleep(0.4)


_, st2 = ops.dw_fwd(xH, D, K, 1, False, epi_alpha=a)
time.sleep(0.5)
if LABEL == "f32":
    case("K1", lambda: ops.pw_gemm(w1, xB, H, B, K, epi_alpha=a))
    case("K3", lambda: ops.pw_gemm(w2, xH, B, H, K, pro=(st2, g, b, a), residual=xB))
    case("B1", lambda: ops.pw_dgrad_gln(w2, xB, H, B, K, xH, g, a, ms))
    case("plain", lambda: ops.pw_gemm(w1, xB, H, B, K))
    case("wgrad1", lambda: ops.pw_wgrad(xH, xB, H, B, K))
    case("wgrad2 pro", lambda: ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms)))
else:
    ops.set_gemm_mode("x6")
    p1 = ops._split_planes(w1, H, B, False)
    p2 = ops._split_planes(w2, B, H, False)
    p2t = ops._split_planes(w2, B, H, True)
    outH = torch.empty(M, H, Kp, device=dev)
    outB = torch.empty(M, B, Kp, device=dev)
    part = torch.empty((M, ctn.lib.ctn_pw_stats_parts(M, H, Kp), 2), dtype=torch.float64, device=dev)
    s = ops._stream()
    case("K1", lambda: ctn.lib.call("ctn_pw_gemm_x6", _p(p1), _p(xB), _p(outH), M, H, B, K, Kp, None, 0, None, None, None, None,
                                    None, _p(a), _p(part), 0, s))
    case("K3", lambda: ctn.lib.call("ctn_pw_gemm_x6", _p(p2), _p(xH), _p(outB), M, B, H, K, Kp, _p(st2), st2.shape[1], _p(g), _p(b),
                                    _p(a), None, _p(xB), None, None, 0, s))
    case("B1", lambda: ctn.lib.call("ctn_pw_dgrad_gln_x6", _p(p2t), _p(xB), _p(outH), M, H, B, K, Kp, _p(xH), _p(g), _p(a), _p(ms),
                                    _p(part), s))
    case("plain", lambda: ctn.lib.call("ctn_pw_gemm_x6", _p(p1), _p(xB), _p(outH), M, H, B, K, Kp, None, 0, None, None, None, None,
                                       None, None, None, 0, s))
    case("wgrad1", lambda: ops.pw_wgrad(xH, xB, H, B, K))
    case("wgrad2 pro", lambda: ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms)))

from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402
m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(dev)
opt = FlatAdam(m.parameters(), lr=1e-3)
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(dev), lens.to(dev), src.to(dev)


def step():
    opt.zero_grad()
    loss = ctn.cal_loss(src, m(mix), lens)[0]
    loss.backward()
    opt.step(max_grad_norm=5.0)
    return loss


print("# first-step loss %s %.6f" % (LABEL, float(step())), flush=True)
case("training step", step)
