#!/usr/bin/env python
"""Record every GEMM call of one training step of a golden model under both arithmetics and print, call by call, how far
the b3 result is from the fp32 one (inputs included) -- finds the first form that deviates by more than arithmetic noise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402
from conftest import load_golden  # noqa: E402

DEV = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "model_tiny_bn"
norm = sys.argv[2] if len(sys.argv) > 2 else "BN"
ops._COMPOSITE = False
rec = []
orig = {k: getattr(ops, k) for k in ("pw_gemm", "pw_wgrad", "pw_dgrad_gln")}


def wrap(k):
    def f(*a, **kw):
        out = orig[k](*a, **kw)
        o = out[0] if isinstance(out, tuple) else out
        desc = "%s R=%s Cn=%s K=%s %s" % (k, a[2], a[3], a[4], {kk: (vv if isinstance(vv, bool) else "set") for kk, vv in kw.items() if vv is not None and kk != "out"})
        ins = [t.detach().clone() for t in a[:2]]
        rec[-1].append((desc, ins, o.detach().clone()))
        return out
    return f


for k in orig:
    setattr(ops, k, wrap(k))


def run(arith):
    ctn.lib.call("ctn_tune", b"arith", arith)
    ops._ws_cache.clear()
    rec.append([])
    gd = load_golden(name)
    N, L, B, H, P, X, R, C = [int(v) for v in gd["cfg"]]
    m = ctn.ConvTasNet(N, L, B, H, P, X, R, C, norm_type=norm, causal=bool(int(gd["causal"])) if "causal" in gd else False)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in gd.items() if k.startswith("p0:")})
    m = m.to(DEV).train()
    mix, src, lens = (torch.from_numpy(gd[k]).to(DEV) for k in ("mixture", "source", "lengths"))
    loss = ctn.cal_loss(src, m(mix), lens)[0]
    loss.backward()
    torch.cuda.synchronize()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


run(0)
run(1)
for (d0, i0, o0), (d1, i1, o1) in zip(rec[0], rec[1]):
    print("%-70s in %.1e %.1e  out %.2e  shape %s" % (d0, rel(i1[0], i0[0]), rel(i1[1], i0[1]), rel(o1, o0), tuple(o0.shape)))
