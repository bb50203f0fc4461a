#!/usr/bin/env python
"""A/B of ctn_tune keys on the split-bf16 weight-gradient kernel, sustained: usage wgrad_ab.py key v0 v1 ..."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
sys.argv = [sys.argv[0], "W1", "0"]
import torch  # noqa: E402
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "b3_only.py")).read().split("fn = fns[form]")[0]
ns = {"__file__": os.path.join(here, "b3_only.py")}
exec(compile(src, "b3_only_setup", "exec"), ns)
ctn, ops = ns["ctn"], ns["ops"]
for rep in range(2):
    for v in vals:
        ctn.lib.call("ctn_tune", key.encode(), v)
        ops._ws_cache.clear()
        for form in ("W1", "W2"):
            fn = ns["fns"][form]
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n, t0 = 0, time.time()
            while time.time() - t0 < 1.0:
                for _ in range(100):
                    fn()
                torch.cuda.synchronize()
                n += 100
            print("%s=%d %s %.2f us" % (key, v, form, (time.time() - t0) / n * 1e6), flush=True)
