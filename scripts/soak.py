#!/usr/bin/env python3
"""Soak run: N training steps of a benchmark workload on fixed synthetic data (the loss must fall and stay finite,
device memory must stop growing after the first steps).    python scripts/soak.py [resnet50|gpt2] [steps] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.optim import FusedSGD
from distributedpytorch_amd.parallel.dp import DataParallel

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda")
torch.manual_seed(0)
if name == "gpt2":
    from distributedpytorch_amd.models import gpt2
    m = gpt2.gpt2_small().to(dev).train()
    x = torch.randint(0, 50257, (batch or 8, 1024), device=dev); y = torch.randint(0, 50257, (batch or 8, 1024), device=dev)
    lr = 0.01
else:
    from distributedpytorch_amd.models import resnet
    m = resnet.resnet50().to(dev).train()
    x = torch.randn(batch or 64, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (batch or 64,), device=dev)
    lr = 0.02
dp = DataParallel(m)
opt = FusedSGD(dp.flat, lr=lr, momentum=0.9, weight_decay=1e-4)
overlap = opt.overlap_with_backward(dp, enabled=(name == "gpt2"))
first = mem0 = None
for i in range(steps):
    loss = m.loss(x, y)
    loss.backward()
    dp.finish()
    opt.step()
    if i % 50 == 0 or i == steps - 1:
        l = float(loss.detach())
        mem = torch.cuda.memory_allocated() / 2 ** 20
        first = l if first is None else first
        mem0 = mem if i == 50 else mem0
        print(f"step {i:4d}  loss {l:8.4f}  allocated {mem:9.1f} MiB  reserved {torch.cuda.memory_reserved() / 2 ** 20:9.1f} MiB", flush=True)
        assert l == l and abs(l) < 1e4, "loss is not finite"
assert l < first, (l, first)
assert mem0 is None or mem <= mem0 * 1.02 + 16, ("device memory keeps growing", mem0, mem)
print(f"soak {name}: ok (overlapped optimizer: {overlap}); loss {first:.4f} -> {l:.4f}")
