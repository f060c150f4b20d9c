#!/usr/bin/env python
"""Weight gradients of a GPT-2 block (fc2, fc, proj, qkv over TOKENS tokens): one launch of csrc/wgrad_wide.hip for the
four layers against four launches of csrc/conv_wgrad.hip, interleaved in one process on the same operands."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402
from bench_conv import timeit  # noqa: E402

C = _ext.load()
M = int(os.environ.get("TOKENS", "16384"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))
LAYERS = [("fc2", 768, 3072), ("fc", 3072, 768), ("proj", 768, 768), ("qkv", 2304, 768)]
if os.environ.get("LMHEAD"):
    LAYERS = [("lm_head", 50304, 768)]
xs = [torch.randn(M, k, device="cuda").to(torch.bfloat16) for _, n, k in LAYERS]
dys = [(torch.randn(M, n, device="cuda") * 0.1).to(torch.bfloat16) for _, n, k in LAYERS]
dws = [torch.zeros(n, k, device="cuda") for _, n, k in LAYERS]
dbs = [torch.zeros(n, device="cuda") for _, n, k in LAYERS]
fl = sum(2.0 * M * n * k for _, n, k in LAYERS)


def group():
    C.wgrad_wide_group(xs, dys, dws, dbs)


def each_wide():
    for q in range(len(LAYERS)):
        C.wgrad_wide_group([xs[q]], [dys[q]], [dws[q]], [dbs[q]])


def narrow():
    for q, (_, n, k) in enumerate(LAYERS):
        C.conv_wgrad(xs[q].view(M, 1, 1, k), dys[q].view(M, 1, 1, n), dws[q].view(n, 1, 1, k), 1, 0, dbs[q])


arms = {"wide, one launch": group, "wide, per layer": each_wide, "conv_wgrad x%d" % len(LAYERS): narrow}
for s in [int(v) for v in os.environ.get("SPLITS", "").split(",") if v]:
    def forced(s=s):
        C.set_wgrad_wide_splitk(s); group(); C.set_wgrad_wide_splitk(0)
    arms[f"wide, one launch, split {s}"] = forced
times = {k: [] for k in arms}
for _ in range(ROUNDS):
    for k, fn in arms.items():
        times[k].append(timeit(fn, 10))
for k in arms:
    med = statistics.median(times[k])
    print(f"tokens {M}  {k:32s} {med:8.1f} us  {fl / med / 1e6:6.0f} TFLOP/s", flush=True)
