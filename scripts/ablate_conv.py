#!/usr/bin/env python
"""Ablation timings of single conv launches (stats on/off, fprop/dgrad) for chosen shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402
from bench_conv import timeit  # noqa: E402

C = _ext.load()
N = 256
for name, H, Cin, K, R, s, p in [("l1.c3 64-256", 56, 64, 256, 1, 1, 0), ("l1.c1 256-64", 56, 256, 64, 1, 1, 0),
                                 ("l2.c3 128-512", 28, 128, 512, 1, 1, 0), ("l1.c2 3x3 64", 56, 64, 64, 3, 1, 1)]:
    Ho = (H + 2 * p - R) // s + 1
    x = torch.randn(N, H, H, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(K, R, R, Cin, device="cuda") * 0.05).to(torch.bfloat16)
    sbuf = torch.zeros(32, 2, K, device="cuda")
    t1 = timeit(lambda: C.conv_fprop(x, w, None, s, p, 0, True, sbuf), 10)
    t0 = timeit(lambda: C.conv_fprop(x, w, None, s, p, 0, False, None), 10)
    y = torch.empty(N, Ho, Ho, K, device="cuda", dtype=torch.bfloat16)
    tc = timeit(lambda: y.copy_(y), 10) if False else 0
    a = torch.empty_like(y)
    tcopy = timeit(lambda: a.copy_(y), 10)
    print(f"{name:16s} fprop stats {t1:7.1f} us  nostats {t0:7.1f} us   copy of y (r+w {2*y.numel()*2/1e6:.0f} MB) {tcopy:7.1f} us", flush=True)
