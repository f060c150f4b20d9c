#!/usr/bin/env python3
"""Per-step wall times of the ResNet-50 step inside ONE process (a host sync after every step), to tell a slow PROCESS
from slow STEPS:   python scripts/slowmode_probe.py [batch] [steps]"""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd import comm as dcomm
from distributedpytorch_amd.models import resnet
from distributedpytorch_amd.optim import FusedSGD
from distributedpytorch_amd.parallel.dp import DataParallel

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
m = resnet.resnet50().to(dev).train()
x = torch.randn(batch, 3, 224, 224, device=dev)
y = torch.randint(0, 1000, (batch,), device=dev)
dp = DataParallel(m, dcomm.Communicator(device=dev))
opt = FusedSGD(dp.flat, lr=0.1, momentum=0.9, weight_decay=1e-4)
opt.grad_scale = dp.grad_scale
ts = []
for s in range(steps + 5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h0 = time.process_time()
    m.loss(x, y).backward()
    dp.finish()
    opt.step()
    h1 = time.process_time()
    torch.cuda.synchronize()
    if s >= 5:
        ts.append(((time.perf_counter() - t0) * 1e3, (h1 - h0) * 1e3))
w = sorted(t for t, _ in ts)
print(f"batch {batch}: wall ms/step min {w[0]:.1f} median {w[len(w) // 2]:.1f} max {w[-1]:.1f}; host cpu ms/step median "
      f"{sorted(h for _, h in ts)[len(ts) // 2]:.1f}; slow steps (> 1.5 x median): {sum(1 for t in w if t > 1.5 * w[len(w) // 2])} of {len(w)}", flush=True)
if w[-1] > 1.5 * w[len(w) // 2]:
    print("   per-step:", " ".join(f"{t:.0f}" for t, _ in ts), flush=True)
