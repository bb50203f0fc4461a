#!/usr/bin/env python
"""Package power of the paper-config training step under the three GEMM arithmetics (see power_lab.sh step); POWER_CONFIG=causal: the
causal cLN config under h3 at the three fusion levels of its channel-wise norms (ctn_tune("cln_fuse", 0 | 1 | 2))."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402

dev = "cuda:0"
SECONDS = float(os.environ.get("SECONDS_PER_CASE", "4"))
CAUSAL = os.environ.get("POWER_CONFIG", "paper") == "causal"
m = (ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2, norm_type="cLN", causal=True) if CAUSAL else ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2)).to(dev)
opt = FlatAdam(m.parameters(), lr=1e-3)
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(dev), lens.to(dev), src.to(dev)


def step():
    opt.zero_grad()
    ctn.cal_loss(src, m(mix), lens)[0].backward()
    opt.step(max_grad_norm=5.0)


def case(name, fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.time()
    while time.time() - t0 < SECONDS:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    t1 = time.time()
    print("case %-28s %.3f %.3f %d %.2f" % (name, t0, t1, n, (t1 - t0) / n * 1e6), flush=True)
    time.sleep(0.5)


case("idle", lambda: time.sleep(0.01))
if CAUSAL:
    for level in (0, 1, 2, 0, 2):
        ctn.lib.call("ctn_tune", b"cln_fuse", level)
        ctn.ops._ws_cache.clear()
        case("causal_step_h3_cln_fuse_%d" % level, step)
    ctn.lib.call("ctn_tune", b"cln_fuse", 2)
else:
    for arith in ("h3", "b6", "fp32", "h3"):
        ctn.set_gemm_arith(arith)
        case("training_step_" + arith, step)
