#!/usr/bin/env python
"""Energy per launch of the block kernels: each case loops for SECONDS while a shell loop samples `rocm-smi --showpower`
(benchmarks/power_lab.sh starts the sampler before this process touches the GPU and joins the two logs).
Prints 'case <name> <t_start> <t_end> <launches> <us_per_launch>' lines; timestamps are time.time()."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

dev = "cuda:0"
M, K = 8, 3199
Kp = ops.padded_frames(K)
SECONDS = float(os.environ.get("SECONDS_PER_CASE", "2.5"))
B, H = 256, 512
torch.manual_seed(0)
xB = torch.randn(M, B, Kp, device=dev); xB[..., K:] = 0
xH = torch.randn(M, H, Kp, device=dev); xH[..., K:] = 0
xH2 = torch.randn(M, H, Kp, device=dev); xH2[..., K:] = 0
w1 = torch.randn(H, B, device=dev) * 0.05
w2 = torch.randn(B, H, device=dev) * 0.05
w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
a = torch.full((1,), 0.25, device=dev)
g = torch.randn(1, H, 1, device=dev)
b = torch.randn(1, H, 1, device=dev)
D = torch.randn(H, 1, 3, device=dev)
ms = torch.tensor([[0.1, 1.3]] * M, device=dev)


def pk(v):
    ctn.lib.call("ctn_tune", b"pk", v)
    ops._ws_cache.clear()


def case(name, fn, setup=None):
    if setup:
        setup()
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
    print("case %-28s %.3f %.3f %d %.2f" % (name.replace(" ", "_"), t0, t1, n, (t1 - t0) / n * 1e6), flush=True)
    time.sleep(0.4)


pk(0)
_, st = ops.pw_gemm(w1, xB, H, B, K, epi_alpha=a)
_, st2 = ops.dw_fwd(xH, D, K, 1, False, epi_alpha=a)
_, s2p = ops.pw_dgrad_gln(w2, xB, H, B, K, xH, g, a, ms)
time.sleep(1.0)
case("idle", lambda: time.sleep(0.01))
case("K1 old", lambda: ops.pw_gemm(w1, xB, H, B, K, epi_alpha=a))
case("K3 old", lambda: ops.pw_gemm(w2, xH, B, H, K, pro=(st2, g, b, a), residual=xB))
case("B1 old", lambda: ops.pw_dgrad_gln(w2, xB, H, B, K, xH, g, a, ms))
case("B5 old", lambda: ops.pw_gemm(w1, xH, B, H, K, trans_w=True, residual=xB))
case("plain old", lambda: ops.pw_gemm(w1, xB, H, B, K))
case("wgrad1 w4", lambda: ops.pw_wgrad(xH, xB, H, B, K))
case("wgrad2 w4 pro", lambda: ops.pw_wgrad(xB, xH, B, H, K, pro=(g, b, a, ms)))
case("dw_fwd", lambda: ops.dw_fwd(xH, D, K, 4, False, pro=(st, g, b, a), epi_alpha=a))
pc = torch.empty((8, M, H), device=dev)
s1p = torch.empty((M, H, 2), dtype=torch.float64, device=dev)
dn1 = torch.empty((M, H, Kp), device=dev)
da1p = torch.empty((M * H,), device=dev)
case("dw_bwd", lambda: ctn.lib.call("ctn_dw_bwd", xH.data_ptr(), xH2.data_ptr(), xH.data_ptr(), dn1.data_ptr(), D.data_ptr(), M, H, K, Kp, 3, 4, 0, 1,
                                   g.data_ptr(), b.data_ptr(), a.data_ptr(), ms.data_ptr(), g.data_ptr(), a.data_ptr(), ms.data_ptr(),
                                   s2p.data_ptr(), s2p.shape[1], pc.data_ptr(), s1p.data_ptr(), ops._stream()))
case("gln_prelu_bwd", lambda: ctn.lib.call("ctn_gln_prelu_bwd", xH.data_ptr(), xH2.data_ptr(), dn1.data_ptr(), M, H, K, Kp, g.data_ptr(), a.data_ptr(),
                                          ms.data_ptr(), s1p.data_ptr(), H, da1p.data_ptr(), ops._stream()))
pk(1)
_, stp = ops.pw_gemm(w1, xB, H, B, K, epi_alpha=a)
case("K1t pk", lambda: ops.pw_gemm(w1t, xB, H, B, K, trans_w=True, epi_alpha=a))
case("K3t pk", lambda: ops.pw_gemm(w2t, xH, B, H, K, trans_w=True, pro=(st2, g, b, a), residual=xB))
case("B1 pk", lambda: ops.pw_dgrad_gln(w2, xB, H, B, K, xH, g, a, ms))
case("B5 pk", lambda: ops.pw_gemm(w1, xH, B, H, K, trans_w=True, residual=xB))
case("plain-t pk", lambda: ops.pw_gemm(w1t, xB, H, B, K, trans_w=True))
pk(0)

# the whole training step
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402
m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(dev)
opt = FlatAdam(m.parameters(), lr=1e-3)
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(dev), lens.to(dev), src.to(dev)


def step():
    opt.zero_grad()
    ctn.cal_loss(src, m(mix), lens)[0].backward()
    opt.step(max_grad_norm=5.0)


def fwd():
    with torch.no_grad():
        m(mix)


case("training step", step)
case("forward only", fwd)
