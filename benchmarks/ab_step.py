#!/usr/bin/env python
"""In-process A/B of library switches on the paper-config training step (one box, one model, interleaved rounds).
usage: python benchmarks/ab_step.py "pk_wgs=4" "pk_wgs=8,wgrad_kernel=0" ...   [env ROUNDS=5 STEPS=8]
Each config is a comma-separated list of ctn_tune keys; prints median / min ms per step per config."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402

configs = sys.argv[1:] or ["pk=1", "pk=0"]
rounds, steps = int(os.environ.get("ROUNDS", "5")), int(os.environ.get("STEPS", "8"))
dev = "cuda:0"
torch.manual_seed(0)
CONFIG = os.environ.get("CONFIG", "paper")          # paper | causal | c3  (bench.py's workloads)
if CONFIG == "causal":
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2, norm_type="cLN", causal=True).to(dev)
    mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
elif CONFIG == "c3":
    m = ctn.ConvTasNet(256, 16, 256, 512, 3, 8, 4, 3).to(dev)
    mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=64000, C=3, sample_rate=16000)))
else:
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(dev)
    mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
opt = FlatAdam(m.parameters(), lr=1e-3)
mix, lens, src = mix.to(dev), lens.to(dev), src.to(dev)


def apply(cfg):
    for kv in cfg.split(","):
        if not kv:
            continue
        k, v = kv.split("=")
        if k == "composite":
            ops._COMPOSITE = bool(int(v))
        elif k == "side":
            ops._SIDE_ENABLED = bool(int(v))
        elif k == "fwd_dual":
            ops._FWD_DUAL = bool(int(v))
        elif k == "cln_side":
            ops._CLN_SIDE = bool(int(v))
        elif k == "small_side":
            ops._SMALL_SIDE = bool(int(v))
        elif k == "hp":                     # 1: run the step's main chain on a HIGH-priority stream (the weight-gradient stream stays normal)
            global HP
            HP = bool(int(v))
        else:
            ctn.lib.call("ctn_tune", k.encode(), int(v))
    ops._ws_cache.clear()


HP = False
_hp_stream = torch.cuda.Stream(device=dev, priority=-1)


def _step():
    opt.zero_grad()
    loss = ctn.cal_loss(src, m(mix), lens)[0]
    loss.backward()
    opt.step(max_grad_norm=5.0)


def step():
    if HP:
        _hp_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(_hp_stream):
            _step()
        torch.cuda.current_stream().wait_stream(_hp_stream)
    else:
        _step()


res = {c: [] for c in configs}
for r in range(rounds):
    for c in configs:
        apply(c)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) / steps * 1e3)
for c in configs:
    v = res[c]
    print("%-44s median %.3f ms  min %.3f  max %.3f  -> %.1f utt/s" % (c, statistics.median(v), min(v), max(v), 8e3 / statistics.median(v)), flush=True)
