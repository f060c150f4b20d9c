#!/usr/bin/env python
"""Stress of tests/test_gpu_e2e.py::test_optimizer_overlapped_with_backward_follows_the_same_trajectory: N rounds of
(plain, overlapped, plain) on tiny_resnet, printing the parameter distances (a race shows as an occasional large one)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models import resnet  # noqa: E402
from distributedpytorch_amd.optim import FusedSGD  # noqa: E402
from distributedpytorch_amd.parallel.dp import DataParallel  # noqa: E402

DEV = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BIG = os.environ.get("BIG", "0") == "1"      # run something heavy first (allocator / clock state as inside the full suite)


def run(overlap):
    torch.manual_seed(7)
    m = resnet.tiny_resnet().to(DEV).train()
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(16, 3, 64, 64, generator=g, device=DEV)
    y = torch.randint(0, 16, (16,), generator=g, device=DEV)
    dp = DataParallel(m, bucket_mb=0.05, first_bucket_mb=0.02)
    opt = FusedSGD(dp.flat, lr=0.05, momentum=0.9, weight_decay=1e-3)
    opt.grad_scale = dp.grad_scale
    opt.overlap_with_backward(dp, enabled=overlap)
    losses = []
    for _ in range(5):
        loss = m.loss(x, y)
        loss.backward()
        dp.finish()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return losses, dp.flat.master.clone()


if BIG:
    big = resnet.resnet50().to(DEV).train()
    xb = torch.randn(64, 3, 224, 224, device=DEV)
    yb = torch.randint(0, 1000, (64,), device=DEV)
    big.loss(xb, yb).backward()
    torch.cuda.synchronize()
    del big, xb, yb
worst = 0.0
for i in range(N):
    l0, p0 = run(False)
    l1, p1 = run(True)
    l2, p2 = run(False)
    d01, d02 = float((p0 - p1).abs().max()), float((p0 - p2).abs().max())
    worst = max(worst, d01)
    flag = "  <-- " if d01 > max(10 * d02, 1e-5) else ""
    print(f"round {i:3d}: plain-vs-overlap {d01:.3e}  plain-vs-plain {d02:.3e}  loss diff {max(abs(a - b) for a, b in zip(l0, l1)):.3e}{flag}", flush=True)
print("worst", worst)
