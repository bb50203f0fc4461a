#!/usr/bin/env python
"""Would two independent half-batch chains (forward + backward + their weight-gradient streams) finish sooner than one full-batch
chain?  Two host threads, each with its own model copy, streams and 4 utterances, against one thread with 8 (paper config,
no optimiser step: forward + loss + backward only).  usage: dual_chain_probe.py [paper|causal]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402

# per-thread workspaces and weight-gradient streams (the library's caches are per device)
_orig_ws = ops._workspace
ops._workspace = lambda nbytes, device, tag: _orig_ws(nbytes, device, (tag, threading.get_ident()))
_tl_side = {}


def _side_stream(device):
    key = (device, threading.get_ident())
    if key not in _tl_side:
        _tl_side[key] = torch.cuda.Stream(device=device)
    return _tl_side[key]


def _join(device=None):
    for (d, t), st in list(_tl_side.items()):
        if t == threading.get_ident() and (device is None or d == device):
            ops._order(st, torch.cuda.current_stream(d))


ops._side_stream = _side_stream
ops.join_side_stream = _join
cfg = sys.argv[1] if len(sys.argv) > 1 else "paper"
dev = "cuda:0"
kw = dict(norm_type="cLN", causal=True) if cfg == "causal" else {}


def make(M):
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2, **kw).to(dev)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    mix, lens, src = next(iter(SyntheticLoader(1, M, samples=32000)))
    return m, opt, mix.to(dev), lens.to(dev), src.to(dev)


def fb(m, opt, mix, lens, src):
    opt.zero_grad()
    ctn.cal_loss(src, m(mix), lens)[0].backward()
    ops.join_side_stream(mix.device)


def run_single(n):
    pack = make(8)
    for _ in range(5):
        fb(*pack)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fb(*pack)
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def run_dual(n):
    packs = [make(4), make(4)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    bar = threading.Barrier(2)

    def worker(i, iters):
        with torch.cuda.stream(streams[i]):
            for _ in range(iters):
                bar.wait()
                fb(*packs[i])

    for iters in (5, n):
        torch.cuda.synchronize()
        t0 = time.time()
        th = [threading.Thread(target=worker, args=(i, iters)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / iters * 1e3
    return dt


n = int(os.environ.get("STEPS", "20"))
a = run_single(n)
print("%s: one chain of 8 utterances        %.3f ms per forward+backward" % (cfg, a), flush=True)
try:
    b = run_dual(n)
    print("%s: two chains of 4 utterances each  %.3f ms per forward+backward (both)" % (cfg, b), flush=True)
except Exception as e:       # the side stream / workspace caches are per device, not per thread
    print("dual run failed:", repr(e))
a = run_single(n)
print("%s: one chain of 8 utterances        %.3f ms (again)" % (cfg, a), flush=True)
