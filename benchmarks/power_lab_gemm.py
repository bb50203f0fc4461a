#!/usr/bin/env python
"""Sustained energy per launch of the split-bf16 GEMM forms (one library per process: CTN_LIB_PATH, LABEL): each case loops for
SECONDS_PER_CASE while benchmarks/power_lab_gemm.sh samples `rocm-smi --showpower`.  FORMS="K1 K3 B1 B5 W1 W2", ARITH=h3|b6|fp32.
Prints 'case <name> <t_start> <t_end> <launches> <us_per_launch>' lines; timestamps are time.time()."""
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
label = os.environ.get("LABEL", "lib")
forms = os.environ.get("FORMS", "K1 K3 B1 B5 W1 W2").split()
seconds = float(os.environ.get("SECONDS_PER_CASE", "2.0"))
if os.environ.get("ARITH"):
    os.environ["CTN_GEMM_ARITH"] = os.environ["ARITH"]
sys.argv = [sys.argv[0], "K1", "0"]
import torch  # noqa: E402
src = open(os.path.join(here, "gemm_only.py")).read().split("fn = fns[form]")[0]
ns = {"__file__": os.path.join(here, "gemm_only.py")}
if os.environ.get("ARITH") == "fp32":       # the fp32-MFMA kernels take the stored matrices, not pieces
    src = src.replace("p1, p2 = _b3_pieces(w1, H, B, False), _b3_pieces(w2, B, H, False)", "p1, p2 = w1, w2")
    src = src.replace("q2, q1 = _b3_pieces(w2, H, B, True), _b3_pieces(w1, B, H, True)", "q2, q1 = w2, w1")
    src = src.replace("M, H, B, K, Kp, 2, None, 0, None, None, None, None, None, _p(a), _p(part), 0, sm)", "M, H, B, K, Kp, 0, None, 0, None, None, None, None, None, _p(a), _p(part), 0, sm)")
    src = src.replace("M, B, H, K, Kp, 2, _p(st2)", "M, B, H, K, Kp, 0, _p(st2)")
    src = src.replace('ctn.lib.call("ctn_pw_dgrad_gln_planes"', 'ctn.lib.call("ctn_pw_dgrad_gln"')
    src = src.replace("M, B, H, K, Kp, 2, None, 0, None, None, None, None, _p(xB), None, None, 0, sm)", "M, B, H, K, Kp, 1, None, 0, None, None, None, None, _p(xB), None, None, 0, sm)")
if os.environ.get("TILE"):                  # tile id of the split-bf16 forward / input-gradient kernels (before the statistics buffers are sized)
    import conv_tasnet_amd as _ctn
    _ctn.lib.call("ctn_tune", b"b3_tile", int(os.environ["TILE"]))
    _ctn.lib.call("ctn_tune", b"b3_tile_k3", int(os.environ["TILE"]))
exec(compile(src, "gemm_only_setup", "exec"), ns)
time.sleep(0.5)
for form in forms:
    fn = ns["fns"][form]
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        n += 100
    t1 = time.time()
    print("case %s_%s %.3f %.3f %d %.2f" % (label, form, t0, t1, n, (t1 - t0) / n * 1e6), flush=True)
    time.sleep(0.3)
