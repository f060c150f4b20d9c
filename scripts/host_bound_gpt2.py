#!/usr/bin/env python3
"""How long does the HOST need to enqueue one GPT-2 small training step (the bench configuration: DataParallel, optimizer
overlapped with backward, grouped wide weight gradients)?  Run at a batch small enough that the GPU finishes first
(2 x 1024 tokens): the wall time per step is then the Python / launch path.  Compare with the GPU's 17 ms at 16 x 1024."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models import gpt2
from distributedpytorch_amd.optim import FusedSGD
from distributedpytorch_amd.parallel.dp import DataParallel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda")
m = gpt2.gpt2_small().to(dev).train()
x = torch.randint(0, 50000, (B, 1024), device=dev)
y = torch.randint(0, 50000, (B, 1024), device=dev)
dp = DataParallel(m)
opt = FusedSGD(dp.flat, lr=0.01, momentum=0.9)
opt.grad_scale = dp.grad_scale
opt.overlap_with_backward(dp, enabled=True)


def step():
    t0 = time.perf_counter()
    loss = m.loss(x, y)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    dp.finish(); opt.step()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


for _ in range(5):
    step()
torch.cuda.synchronize()
acc = [0.0, 0.0, 0.0]
N = 20
t0 = time.perf_counter()
for _ in range(N):
    for i, v in enumerate(step()):
        acc[i] += v
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"batch {B} x 1024: host per step {1e3 * (t1 - t0) / N:.2f} ms (forward {1e3 * acc[0] / N:.2f}, backward {1e3 * acc[1] / N:.2f}, "
      f"finish + optimizer {1e3 * acc[2] / N:.2f}); with the final sync {1e3 * (t2 - t0) / N:.2f} ms per step")
