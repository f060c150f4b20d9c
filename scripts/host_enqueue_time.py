#!/usr/bin/env python3
"""Host-side cost of enqueueing one training step (launch-bound regime: a tiny batch, so the GPU is never the
bottleneck), i.e. the head room between the Python/launch path and the time the GPU needs at the benchmark batch.

    python scripts/host_enqueue_time.py [resnet50|resnet152|gpt2]
Measured (round 2): ResNet-50 7.0 ms (GPU step 19.2 ms), see docs/DESIGN.md for the others."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.optim import FusedSGD
from distributedpytorch_amd.parallel.dp import DataParallel

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
dev = torch.device("cuda")
if name == "gpt2":
    from distributedpytorch_amd.models import gpt2
    m = gpt2.gpt2_small().to(dev).train()
    xs = torch.randint(0, 50000, (1, 128), device=dev); ys = torch.randint(0, 50000, (1, 128), device=dev)
else:
    from distributedpytorch_amd.models import resnet
    m = getattr(resnet, name)().to(dev).train()
    xs = torch.randn(8, 3, 224, 224, device=dev); ys = torch.randint(0, 1000, (8,), device=dev)
dp = DataParallel(m); opt = FusedSGD(dp.flat, lr=0.01, momentum=0.9, weight_decay=1e-4)


def step_small():
    m.loss(xs, ys).backward(); dp.finish(); opt.step()


for _ in range(5):
    step_small()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step_small()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{name}: host enqueue per step {(t1 - t0) / 20 * 1e3:.2f} ms   (with final sync {(t2 - t0) / 20 * 1e3:.2f} ms per step at the tiny batch)")
