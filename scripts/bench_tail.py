"""Fused bottleneck-tail backward (csrc/tail_bwd.hip) against the three kernels it replaces, on the ResNet-50 stage-1
shape (batch 256: 802,816 pixels, 64 -> 256 channels).  Interleaved arms, median of ROUNDS.

    python scripts/bench_tail.py [N] [HW] [CM]        (stage 2: 256 28 128)
"""
import json
import os
import statistics
import sys

import torch

from distributedpytorch_amd.ops import _ext

C = _ext.load()
DEV = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 56
CM = int(sys.argv[3]) if len(sys.argv) > 3 else 64
CO = 4 * CM
ROUNDS = int(os.environ.get("ROUNDS", "20"))
P = N * HW * HW

dout = torch.randn(N, HW, HW, CO, device=DEV).to(torch.bfloat16)
craw = torch.randn(N, HW, HW, CO, device=DEV).to(torch.bfloat16)
xin = torch.randn(N, HW, HW, CM, device=DEV).to(torch.bfloat16)
sx = torch.randn(N, HW, HW, CM, device=DEV).to(torch.bfloat16)
w = (torch.randn(CO, 1, 1, CM, device=DEV) * 0.1).to(torch.bfloat16)
wt = C.transpose_filter(w)
kout = torch.randint(0, 256, (P * CO // 8,), dtype=torch.uint8, device=DEV)
smask = torch.randint(0, 256, (P * CM // 8,), dtype=torch.uint8, device=DEV)
co = torch.rand(5, CO, device=DEV)
dw = torch.zeros(CO, 1, 1, CM, device=DEV)
stats = torch.zeros(32, 2, CM, device=DEV)


def fused():
    return C.tail_bwd(dout, kout, craw, co, xin, wt, dw, sx, smask, stats)


def seq():
    g3 = C.bn_bwd_apply(dout, kout, craw, co, True)
    C.conv_wgrad(xin, g3, dw, 1, 0)
    return C.conv_dgrad(g3, wt, HW, HW, 1, 0, None, None, sx, smask, stats)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for _ in range(3):
    fused(); seq()
torch.cuda.synchronize()
tf, ts = [], []
for _ in range(ROUNDS):
    tf.append(timed(fused)); ts.append(timed(seq))
bytes_fused = P * (CO * 4.125 + CM * 2 * 3 + CM / 8)
out = {"shape": [N, HW, HW, CM, CO], "fused_us": statistics.median(tf), "fused_min_us": min(tf),
       "sequence_us": statistics.median(ts), "sequence_min_us": min(ts),
       "fused_tb_per_s": bytes_fused / statistics.median(tf) / 1e6}
print(json.dumps(out))
