#!/usr/bin/env python
"""Which ATen ops run in a ResNet-50 training step besides our kernels (copies, fills, adds)?"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models import resnet  # noqa: E402
from distributedpytorch_amd.optim import FusedSGD  # noqa: E402
from distributedpytorch_amd.parallel.dp import DataParallel  # noqa: E402

dev = torch.device("cuda")
m = resnet.resnet50().to(dev).train()
x = torch.randn(64, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (64,), device=dev)
dp = DataParallel(m); opt = FusedSGD(dp.flat, lr=0.1, momentum=0.9, weight_decay=1e-4)


def step():
    m.loss(x, y).backward(); dp.finish(); opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(group_by_stack_n=4), key=lambda e: -e.count):
    if e.key.startswith("aten::") and e.count >= 4:
        print(f"{e.key:32s} x{e.count:4d}  cpu {e.cpu_time_total:8.0f} us   {' <- '.join(s.split('/')[-1] for s in e.stack[:3])}")
