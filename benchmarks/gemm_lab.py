#!/usr/bin/env python
"""Every GEMM form of the composite stacks at the paper shapes, alone and sustained (SECONDS_PER_CASE of back-to-back launches each),
for a list of ctn_tune settings.  usage: gemm_lab.py [key=value[,key=value] ...]   e.g.  gemm_lab.py b3_tpw=1 b3_tpw=2 b3_tpw=3
Prints one line per setting: us per launch of K1 K3 B1 B5 W1 W2 (FORMS env selects)."""
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
forms = os.environ.get("FORMS", "K1 K3 B1 B5 W1 W2").split()
seconds = float(os.environ.get("SECONDS_PER_CASE", "1.0"))
settings = [a for a in sys.argv[1:] if "=" in a] or ["b3_tpw=1"]
sys.argv = [sys.argv[0], "K1", "0"]
import torch  # noqa: E402
src = open(os.path.join(here, "gemm_only.py")).read().split("fn = fns[form]")[0]
ns = {"__file__": os.path.join(here, "gemm_only.py")}
exec(compile(src, "gemm_only_setup", "exec"), ns)
ctn = ns["ctn"]
print("arith", ctn.gemm_arith(), "lib", ctn.LIB_PATH, flush=True)
for setting in settings:
    for kv in setting.split(","):
        k, v = kv.split("=")
        ctn.lib.call("ctn_tune", k.encode(), int(v))
    row = []
    for form in forms:
        fn = ns["fns"][form]
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n, t0 = 0, time.time()
        while time.time() - t0 < seconds:
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            n += 200
        row.append("%s %6.2f" % (form, (time.time() - t0) / n * 1e6))
    print("%-28s %s" % (setting, "  ".join(row)), flush=True)
