import os, sys
sys.path.insert(0, os.getcwd())
import torch
import conv_tasnet_amd as ctn
from conv_tasnet_amd.optim import FlatAdam
from conv_tasnet_amd.train import SyntheticLoader
DEV = "cuda:0"
mix, lens, src = next(iter(SyntheticLoader(1, 8, samples=32000)))
mix, lens, src = mix.to(DEV), lens.to(DEV), src.to(DEV)
def run(arith, steps=7):
    ctn.set_gemm_arith(arith)
    torch.manual_seed(0)
    m = ctn.ConvTasNet(256, 20, 256, 512, 3, 8, 4, 2).to(DEV)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    ls = []
    for _ in range(steps):
        opt.zero_grad()
        loss = ctn.cal_loss(src, m(mix), lens)[0]
        loss.backward()
        opt.step(max_grad_norm=5.0)
        ls.append(float(loss.detach()))
    return ls, opt.flat_params.detach().clone()
tag = "dual=%s side=%s" % (os.environ.get("CTN_FWD_DUAL", "1"), os.environ.get("CTN_SIDE_STREAM", "1"))
l32, p32 = run("fp32")
for a in sys.argv[1:] or ["h3", "h3", "b6"]:
    l, p = run(a)
    print(tag, a, "max |loss - fp32| %.2e" % max(abs(x - y) for x, y in zip(l, l32)), "|p - p32| %.4e" % float((p - p32).double().norm()), " ".join("%.6f" % v for v in l), flush=True)
