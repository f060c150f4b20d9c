#!/usr/bin/env python
"""HBM read / write / copy bandwidth on this device (sizes bracketing the 256 MB Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_conv import timeit  # noqa: E402

for mb in (64, 206, 411, 822, 1644):
    n = mb * 1000 * 1000 // 2
    y = torch.randn(n, device="cuda").to(torch.bfloat16)
    a = torch.empty_like(y)
    tw = timeit(lambda: a.zero_(), 10)
    tr = timeit(lambda: y.float().sum() if False else torch.max(y), 10)
    tc = timeit(lambda: a.copy_(y), 10)
    print(f"{mb:5d} MB: write {mb / tw * 1e3 / 1e3:6.2f} TB/s ({tw:7.1f} us)  read(max) {mb / tr * 1e3 / 1e3:6.2f} TB/s ({tr:7.1f} us)  "
          f"copy {2 * mb / tc * 1e3 / 1e3:6.2f} TB/s r+w ({tc:7.1f} us)", flush=True)
