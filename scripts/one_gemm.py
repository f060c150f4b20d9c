#!/usr/bin/env python
"""One GEMM shape through conv_fprop / conv_dgrad (R=S=1) for profilers and quick timings:
    python scripts/one_gemm.py M K N [wide|narrow] [fprop|dgrad] [iters]
prints the mean time per launch (hipEvents) and TFLOP/s on random operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402
from bench_conv import timeit  # noqa: E402

M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
arm = sys.argv[4] if len(sys.argv) > 4 else "wide"
mode = sys.argv[5] if len(sys.argv) > 5 else "fprop"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
C = _ext.load()
C.set_wide_gemm_min_tiles(1 if arm == "wide" else 0)
x = torch.randn(M, 1, 1, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, 1, 1, K, device="cuda") * 0.02).to(torch.bfloat16)
if mode == "fprop":
    fn = lambda: C.conv_fprop(x, w, None, 1, 0, 0, False, None)  # noqa: E731
else:
    fn = lambda: C.conv_dgrad(x, w, 1, 1, 1, 0, None, None)  # noqa: E731
for _ in range(3):
    fn()
t = timeit(fn, iters)
print(f"{arm} {mode} M={M} K={K} N={N}: {t:8.1f} us  {2.0 * M * K * N / t / 1e6:6.0f} TF", flush=True)
