#!/usr/bin/env python
"""Run only the pre-split bf16 GEMM (for rocprofv3 --pmc)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import conv_tasnet_amd as ctn  # noqa: E402
from conv_tasnet_amd import ops  # noqa: E402

M, K, R, Cn = 8, 3199, 512, 256
Kp = ops.padded_frames(K)
lib = ctn.lib
lib.ctn_tune_pw_tile(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
W = torch.randn(R, Cn, device="cuda:0") * 0.05
X = torch.randn(M, Cn, Kp, device="cuda:0")
Wp = ops._split_planes(W, R, Cn, False)
Xp = torch.empty((3,) + tuple(X.shape), dtype=torch.bfloat16, device="cuda:0")
lib.call("ctn_split_act", X.data_ptr(), Xp.data_ptr(), X.numel(), 0)
out = torch.empty(M, R, Kp, device="cuda:0")
for _ in range(12):
    lib.call("ctn_pw_gemm_p6", Wp.data_ptr(), 0, Xp.data_ptr(), out.data_ptr(), 0, M, R, Cn, K, Kp, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
torch.cuda.synchronize()
