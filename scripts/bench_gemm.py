#!/usr/bin/env python
"""GEMM shapes of GPT-2 small (tokens = 8192) and the large 1x1 layers of ResNet-50 through the conv kernels as
R=S=1 convolutions: the 256 x 256 main loop (csrc/gemm_wide.hip) against the 128 x 128 one (csrc/conv_igemm.hip)
and, with WITH_LIB=1, against the library GEMM (hipBLASLt through torch.matmul) - comparison line only, nothing in
the framework calls it.  All arms run interleaved in ONE process on the same random operands (several rounds,
median), since separate processes differ by more than the kernels do."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402
from bench_conv import timeit  # noqa: E402

C = _ext.load()
M = int(os.environ.get("TOKENS", "8192"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))
WITH_LIB = os.environ.get("WITH_LIB", "0") == "1"
SHAPES = [("qkv 768->2304", M, 768, 2304), ("proj 768->768", M, 768, 768), ("fc 768->3072", M, 768, 3072),
          ("fc2 3072->768", M, 3072, 768), ("lm_head 768->50304", M, 768, 50304),
          ("r50 l3.c3 256->1024", 50176, 256, 1024), ("r50 l3.c1 1024->256", 50176, 1024, 256),
          ("r50 l2.c3 128->512", 200704, 128, 512), ("r50 l4.c3 512->2048", 12544, 512, 2048)]
only = os.environ.get("ONLY")
rows = []
for name, m, K_in, N_out in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn(m, 1, 1, K_in, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N_out, 1, 1, K_in, device="cuda") * 0.02).to(torch.bfloat16)
    dy = torch.randn(m, 1, 1, N_out, device="cuda").to(torch.bfloat16)
    wt = C.transpose_filter(w)
    fl = 2.0 * m * K_in * N_out
    arms = {"wide fprop": (1, lambda: C.conv_fprop(x, w, None, 1, 0, 0, False, None)),
            "128  fprop": (0, lambda: C.conv_fprop(x, w, None, 1, 0, 0, False, None)),
            "wide dgrad": (1, lambda: C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None)),
            "128  dgrad": (0, lambda: C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None))}
    if os.environ.get("EPI") == "1":
        # the fused epilogues GPT-2 uses on these shapes (wide kernel): bias / bias + residual / bias + GELU twin / GELU backward
        bias = torch.randn(N_out, device="cuda")
        res = torch.randn(m, 1, 1, N_out, device="cuda").to(torch.bfloat16)
        h = torch.randn(m, 1, 1, K_in, device="cuda").to(torch.bfloat16)
        arms = {"wide plain": (1, lambda: C.conv_fprop(x, w, None, 1, 0, 0, False, None)),
                "wide bias": (1, lambda: C.conv_fprop(x, w, bias, 1, 0, 0, False, None)),
                "wide bias+res": (1, lambda: C.conv_fprop(x, w, bias, 1, 0, 0, False, None, res)),
                "wide bias+twin": (1, lambda: C.conv_fprop(x, w, bias, 1, 0, 0, False, None, None, True)),
                "wide dgrad": (1, lambda: C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None)),
                "wide dgrad*gelu'": (1, lambda: C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None, None, None, None, h)),
                "gelu kernel (N)": (1, lambda: C.ew2(res, res, 3))}
    if WITH_LIB:
        x2, w2, dy2 = x.view(m, K_in), w.view(N_out, K_in), dy.view(m, N_out)
        arms["lib  fprop"] = (0, lambda: torch.matmul(x2, w2.t()))
        arms["lib  dgrad"] = (0, lambda: torch.matmul(dy2, w2))
    times = {k: [] for k in arms}
    for _ in range(ROUNDS):
        for k, (mt, fn) in arms.items():
            C.set_wide_gemm_min_tiles(mt)
            times[k].append(timeit(fn, 10))
    C.set_wide_gemm_min_tiles(1)
    row = {"name": name, "M": m, "K": K_in, "N": N_out}
    line = f"{name:22s}"
    for k in arms:
        med, best = statistics.median(times[k]), min(times[k])
        row[k] = {"median_us": med, "min_us": best, "tflops_median": fl / med / 1e6}
        line += f" | {k} {med:7.1f} us {fl / med / 1e6:5.0f} TF"
    rows.append(row)
    print(line, flush=True)
out = os.environ.get("OUT")
if out:
    with open(out, "w") as f:
        json.dump(rows, f, indent=1)
