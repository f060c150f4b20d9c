#!/usr/bin/env python3
"""Launch ONE conv kernel config a few times (target for rocprofv3 --pmc runs).
   usage: one_conv.py H C K R stride pad {fprop|dgrad|wgrad} [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402

H, Cin, K, R, s, p = (int(v) for v in sys.argv[1:7])
what = sys.argv[7]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
C = _ext.load()
N = int(os.environ.get("BATCH", "256"))
Ho = (H + 2 * p - R) // s + 1
x = torch.randn(N, H, H, Cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(K, R, R, Cin, device="cuda") * 0.05).to(torch.bfloat16)
dy = torch.randn(N, Ho, Ho, K, device="cuda").to(torch.bfloat16)
wt = C.transpose_filter(w)
dw = torch.zeros(K, R, R, Cin, device="cuda")
sbuf = torch.zeros(32, 2, K, device="cuda")
torch.cuda.synchronize()
for _ in range(iters):
    if what == "fprop":
        C.conv_fprop(x, w, None, s, p, 0, True, sbuf)
    elif what == "dgrad":
        C.conv_dgrad(dy, wt, H, H, s, p, None, None)
    else:
        C.conv_wgrad(x, dy, dw, s, p)
torch.cuda.synchronize()
