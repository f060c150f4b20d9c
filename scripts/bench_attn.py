#!/usr/bin/env python3
"""Causal attention kernels (D=64) at the GPT-2 small shape: time and TFLOP/s of fwd / bwd, kernel by kernel.

    python scripts/bench_attn.py [B] [T] [H]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 12
C = _ext.load()
qkv = (torch.randn(B, T, 3, H, 64, device="cuda") * 0.5).to(torch.bfloat16)
dout = torch.randn(B, T, H, 64, device="cuda").to(torch.bfloat16)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


out, lse = C.attention_fwd(qkv)
mm = 2.0 * B * H * T * T * 64 * 0.5          # one causal T x T x 64 product
t_f = timeit(lambda: C.attention_fwd(qkv))
t_b = timeit(lambda: C.attention_bwd(qkv, out, dout, lse))
print(f"attention B={B} T={T} H={H} D=64   fwd {t_f:7.1f} us {2 * mm / t_f / 1e6:6.0f} TF | bwd (dQ with delta + dK/dV) {t_b:7.1f} us "
      f"{7 * mm / t_b / 1e6:6.0f} TF (4 + 3 products)")
