#!/usr/bin/env python3
"""cProfile of the host side of a training step (where do the ~11 ms of Python per GPT-2 step go?):
    python scripts/host_profile.py [gpt2|resnet50] [steps]"""
import cProfile
import os
import pstats
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd import comm as dcomm
from distributedpytorch_amd.optim import FusedSGD
from distributedpytorch_amd.parallel.dp import DataParallel

which = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
if which == "gpt2":
    from distributedpytorch_amd.data import synthetic_batch
    from distributedpytorch_amd.models import gpt2
    m = gpt2.gpt2_small().to(dev).train()
    x, y = synthetic_batch("tokens", 16, dev, seed=1, seq_len=1024)
else:
    from distributedpytorch_amd.models import resnet
    m = resnet.resnet50().to(dev).train()
    x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
dp = DataParallel(m, dcomm.Communicator(device=dev))
opt = FusedSGD(dp.flat, lr=0.01, momentum=0.9)
opt.grad_scale = dp.grad_scale
if which == "gpt2":
    opt.overlap_with_backward(dp, enabled=True)


def step():
    m.loss(x, y).backward()
    dp.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
# backward on the calling thread, so that the Python backward functions show up in the profile (the engine's device thread
# is invisible to cProfile)
with torch.autograd.set_multithreading_enabled(False):
    pr.enable()
    for _ in range(steps):
        step()
        torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
