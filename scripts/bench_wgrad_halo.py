#!/usr/bin/env python3
"""3 x 3 weight gradient: correctness against an fp32 reference and time per launch.
   DPA_WGRAD_HALO=0|1 selects conv_wgrad.hip's implicit GEMM or conv_wgrad_halo.hip (read once per process).
   usage: bench_wgrad_halo.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402

C = _ext.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
print("DPA_WGRAD_HALO =", os.environ.get("DPA_WGRAD_HALO", "(default)"), flush=True)
for (H, Cin, K) in ((56, 64, 64), (28, 128, 128), (56, 128, 64), (28, 64, 192)):
    torch.manual_seed(0)
    n = 8
    x = torch.randn(n, H, H, Cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(n, H, H, K, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(K, 3, 3, Cin, device="cuda")
    C.conv_wgrad(x, dy, dw, 1, 1)
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, Cin, 3, 3), dy.float().permute(0, 3, 1, 2),
                                      stride=1, padding=1).permute(0, 2, 3, 1)
    err = float((dw - ref).abs().max() / ref.abs().max())
    print(f"H={H} C={Cin} K={K}: max err / max |ref| = {err:.2e}", flush=True)
    assert err < 2e-3, err
for (H, Cin, K) in ((56, 64, 64), (28, 128, 128)):
    x = torch.randn(N, H, H, Cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, H, H, K, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(K, 3, 3, Cin, device="cuda")
    for _ in range(3):
        C.conv_wgrad(x, dy, dw, 1, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        C.conv_wgrad(x, dy, dw, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    fl = 2.0 * N * H * H * K * Cin * 9
    print(f"batch {N} H={H} C={Cin} K={K}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
