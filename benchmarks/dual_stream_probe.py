#!/usr/bin/env python
"""Would two half-batch chains on two streams beat one full-batch chain?  Timing probe only (gradients of the two halves
overwrite each other in the flat buffer, which does not change the work done)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conv_tasnet_amd.optim import FlatAdam  # noqa: E402
from conv_tasnet_amd.train import SyntheticLoader  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(dev)
opt = FlatAdam(m.parameters(), lr=1e-3)
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(dev), lens.to(dev), src.to(dev)

# per-stream workspaces and side streams for the probe
_orig_ws, _orig_side = ops._workspace, ops._side_stream
ops._workspace = lambda nbytes, device, tag: _orig_ws(nbytes, device, (tag, torch.cuda.current_stream().cuda_stream))
_sides = {}


def _side(device):
    k = torch.cuda.current_stream().cuda_stream
    if k not in _sides:
        _sides[k] = torch.cuda.Stream(device=device)
    return _sides[k]


ops._side_stream = _side
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def full():
    opt.zero_grad()
    loss = ctn.cal_loss(src, m(mix), lens)[0]
    loss.backward()
    opt.step(max_grad_norm=5.0)


def halves(fwd_only=False):
    opt.zero_grad()
    cur = torch.cuda.current_stream()
    for st, sl in ((s1, slice(0, 4)), (s2, slice(4, 8))):
        st.wait_stream(cur)
    losses = []
    for st, sl in ((s1, slice(0, 4)), (s2, slice(4, 8))):
        with torch.cuda.stream(st):
            losses.append(ctn.cal_loss(src[sl], m(mix[sl]), lens[sl])[0])
    for st, l in zip((s1, s2), losses):
        with torch.cuda.stream(st):
            opt._written.clear()            # timing probe: the second half overwrites the first half's gradients
            l.backward()
    for st in (s1, s2):
        cur.wait_stream(st)
    opt.step(max_grad_norm=5.0)


def fwd_full():
    with torch.no_grad():
        m(mix)


def fwd_halves():
    cur = torch.cuda.current_stream()
    with torch.no_grad():
        for st, sl in ((s1, slice(0, 4)), (s2, slice(4, 8))):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                m(mix[sl])
        for st in (s1, s2):
            cur.wait_stream(st)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, fn in (("full step", full), ("two half-batch chains", halves), ("forward full", fwd_full), ("forward two halves", fwd_halves)):
    print("%-24s %.2f ms" % (name, timeit(fn)), flush=True)
