#!/usr/bin/env python
"""conv3 + bn3 of a bottleneck: stored (conv + BN kernels) vs recomputed (BN fused into conv epilogues)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402
from bench_conv import timeit  # noqa: E402

C = _ext.load()
N = 256
for name, H, Cin, K in [("l1.c3", 56, 64, 256), ("l2.c3", 28, 128, 512), ("l3.c3", 14, 256, 1024)]:
    a2 = torch.randn(N, H, H, Cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(K, 1, 1, Cin, device="cuda") * 0.05).to(torch.bfloat16)
    idn = torch.randn(N, H, H, K, device="cuda").to(torch.bfloat16)
    dout = torch.randn(N, H, H, K, device="cuda").to(torch.bfloat16)
    sb = torch.zeros(32, 2, K, device="cuda"); sb2 = torch.zeros(32, 2, K, device="cuda")
    gamma = torch.ones(K, device="cuda"); beta = torch.zeros(K, device="cuda")
    P = N * H * H
    # stored path
    t_conv = timeit(lambda: C.conv_fprop(a2, w, None, 1, 0, 0, True, sb), 10)
    c3, _ = C.conv_fprop(a2, w, None, 1, 0, 0, True, sb)
    mean, invstd, scale, shift = C.bn_finalize(sb, P, gamma, beta, None, None, 0.1, 1e-5, True)
    t_apply = timeit(lambda: C.bn_apply(c3, idn, scale, shift, True, True), 10)
    out, mask = C.bn_apply(c3, idn, scale, shift, True, True)
    t_bwd = timeit(lambda: C.bn_backward(dout, mask, c3, gamma, mean, invstd, True, None, None, sb2), 10)
    # recompute path
    t_s = timeit(lambda: C.conv_fprop_stats(a2, w, 1, 0, sb), 10); sb.zero_()
    t_f = timeit(lambda: C.conv_bn_res_relu(a2, w, 1, 0, scale, shift, idn), 10)
    t_bs = timeit(lambda: C.conv_bnb_stats(a2, w, 1, 0, dout, mask, sb2), 10)
    co = C.bn_bwd_coefs(sb2, P, gamma, mean, invstd, None, None)
    t_ba = timeit(lambda: C.conv_bnb_apply(a2, w, 1, 0, dout, mask, co[2], co[3], co[4]), 10)
    print(f"{name}: stored fwd {t_conv:.0f}+{t_apply:.0f}={t_conv + t_apply:.0f} us, bwd {t_bwd:.0f} us | "
          f"recompute fwd {t_s:.0f}+{t_f:.0f}={t_s + t_f:.0f} us, bwd {t_bs:.0f}+{t_ba:.0f}={t_bs + t_ba:.0f} us", flush=True)
