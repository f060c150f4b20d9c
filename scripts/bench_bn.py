#!/usr/bin/env python3
"""BatchNorm forward-apply / backward (stats + apply) at the ResNet-50 shapes: time and effective HBM rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402

C = _ext.load()
N = int(os.environ.get("BATCH", "256"))


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


print(f"{'shape':22} {'MB/tensor':>9} | apply+relu us  TB/s | apply+res+relu us  TB/s | bwd(relu) us  TB/s")
for name, hw, ch in [("l1 64", 56, 64), ("l1 256", 56, 256), ("l2 128", 28, 128), ("l2 512", 28, 512), ("l3 256", 14, 256),
                     ("l3 1024", 14, 1024), ("l4 512", 7, 512), ("l4 2048", 7, 2048)]:
    x = torch.randn(N, hw, hw, ch, device="cuda").to(torch.bfloat16)
    res = torch.randn(N, hw, hw, ch, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, hw, hw, ch, device="cuda").to(torch.bfloat16)
    scale = torch.rand(ch, device="cuda") + 0.5
    shift = torch.randn(ch, device="cuda")
    gamma = torch.rand(ch, device="cuda") + 0.5
    mean = torch.zeros(ch, device="cuda")
    invstd = torch.ones(ch, device="cuda")
    mb = x.numel() * 2 / 1e6
    y, mask = C.bn_apply(x, None, scale, shift, True, True)
    t1 = timeit(lambda: C.bn_apply(x, None, scale, shift, True, True))
    t2 = timeit(lambda: C.bn_apply(x, res, scale, shift, True, True))
    t3 = timeit(lambda: C.bn_backward(dy, mask, x, gamma, mean, invstd, True, None, None, None))
    print(f"{name + f' {hw}x{hw}':22} {mb:9.1f} | {t1:8.1f} {2 * mb / t1:6.2f} | {t2:8.1f} {3 * mb / t2:6.2f} | {t3:8.1f} {5 * mb / t3:6.2f}")
