#!/usr/bin/env python3
"""Which Python lines still launch stock ATen kernels (fill / copy / elementwise) inside a training step?
Runs 2 profiled ResNet-50 (or --model gpt2 / mnist) steps under torch.profiler with stacks and prints, per ATen op that
launched a device kernel, the innermost frames of this package.    python scripts/find_aten_launches.py [--model M] [--batch B]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--model", default="resnet50")
p.add_argument("--batch", type=int, default=64)
args = p.parse_args()
dev = torch.device("cuda", 0)
from distributedpytorch_amd.optim import FusedSGD  # noqa: E402
from distributedpytorch_amd.parallel.dp import DataParallel  # noqa: E402

torch.manual_seed(0)
if args.model == "gpt2":
    from distributedpytorch_amd.models import gpt2
    model = gpt2.gpt2_small().to(dev).train()
    x = torch.randint(0, 50257, (args.batch, 1024), device=dev); y = torch.randint(0, 50257, (args.batch, 1024), device=dev)
else:
    from distributedpytorch_amd.models import resnet
    model = resnet.resnet50().to(dev).train()
    x = torch.randn(args.batch, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (args.batch,), device=dev)
dp = DataParallel(model)
opt = FusedSGD(dp.flat, lr=0.1, momentum=0.9, weight_decay=1e-4)


def step():
    loss = model.loss(x, y)
    loss.backward()
    dp.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if "distributedpytorch_amd" in f or "bench" in f or "find_aten" in f]
    by[(ev.name, " <- ".join(frames[:3]) or "(no package frame)")] += 1
for (name, where), n in by.most_common(40):
    print(f"{n / 2:6.1f}/step  {name:28s} {where}")
