#!/usr/bin/env python
"""GEMM shapes of GPT-2 small (tokens = 8192) through the conv kernels as R=S=1 convolutions: TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402
from bench_conv import timeit  # noqa: E402

C = _ext.load()
M = int(os.environ.get("TOKENS", "8192"))
for name, K_in, N_out in [("qkv 768->2304", 768, 2304), ("proj 768->768", 768, 768), ("fc 768->3072", 768, 3072),
                          ("fc2 3072->768", 3072, 768), ("lm_head 768->50304", 768, 50304)]:
    x = torch.randn(M, 1, 1, K_in, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N_out, 1, 1, K_in, device="cuda") * 0.02).to(torch.bfloat16)
    dy = torch.randn(M, 1, 1, N_out, device="cuda").to(torch.bfloat16)
    wt = C.transpose_filter(w)
    dw = torch.zeros(N_out, 1, 1, K_in, device="cuda")
    fl = 2.0 * M * K_in * N_out
    tf = timeit(lambda: C.conv_fprop(x, w, None, 1, 0, 0, False, None), 10)
    td = timeit(lambda: C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None), 10)
    tw = timeit(lambda: C.conv_wgrad(x, dy, dw, 1, 0), 10)
    if os.environ.get("WITH_LIB", "0") == "1":      # the library GEMM (hipBLASLt through torch) on the same operands
        x2, w2, dy2 = x.view(M, K_in), w.view(N_out, K_in), dy.view(M, N_out)
        lf = timeit(lambda: torch.matmul(x2, w2.t()), 10)
        ld = timeit(lambda: torch.matmul(dy2, w2), 10)
        lw = timeit(lambda: torch.matmul(dy2.t(), x2), 10)
        print(f"{'  library':20s} fprop {lf:7.1f} us {fl / lf / 1e6:6.0f} TF | dgrad {ld:7.1f} us {fl / ld / 1e6:6.0f} TF | wgrad {lw:7.1f} us {fl / lw / 1e6:6.0f} TF", flush=True)
    print(f"{name:20s} fprop {tf:7.1f} us {fl / tf / 1e6:6.0f} TF | dgrad {td:7.1f} us {fl / td / 1e6:6.0f} TF | wgrad {tw:7.1f} us {fl / tw / 1e6:6.0f} TF", flush=True)
