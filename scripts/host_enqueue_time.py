#!/usr/bin/env python3
"""Host-side cost of enqueueing one ResNet-50 training step (launch-bound regime: batch 8), i.e. the head room
between the Python/launch path and the 22 ms the GPU needs at batch 256.  Measured: 7.0 ms per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models import resnet
from distributedpytorch_amd.optim import FusedSGD
from distributedpytorch_amd.parallel.dp import DataParallel
dev = torch.device("cuda")
m = resnet.resnet50().to(dev).train()
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
dp = DataParallel(m); opt = FusedSGD(dp.flat, lr=0.1, momentum=0.9, weight_decay=1e-4)
def step():
    m.loss(x, y).backward(); dp.finish(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
# host-side enqueue time: small batch so the GPU is never the bottleneck
xs = x[:8].contiguous(); ys = y[:8].contiguous()
def step_small():
    m.loss(xs, ys).backward(); dp.finish(); opt.step()
for _ in range(5): step_small()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step_small()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per step: {(t1-t0)/20*1e3:.2f} ms   (with final sync {(t2-t0)/20*1e3:.2f} ms per step at batch 8)")
