"""LayerNorm kernels (csrc/transformer.hip) at the GPT-2 shape, stand-alone: forward, backward (+ residual addend) and
the partial fold, median us and effective TB/s.

    python scripts/bench_ln.py [ROWS] [C]
"""
import json
import statistics
import sys

import torch

from distributedpytorch_amd.ops import _ext

C_ = _ext.load()
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = "cuda"
x = torch.randn(ROWS, C, device=dev).to(torch.bfloat16)
dy = torch.randn(ROWS, C, device=dev).to(torch.bfloat16)
ad = torch.randn(ROWS, C, device=dev).to(torch.bfloat16)
g = torch.rand(C, device=dev) + 0.5
b = torch.randn(C, device=dev)
dga, dba = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
y, mean, rstd = C_.layernorm_fwd(x, g, b, 1e-5)
part = C_.layernorm_bwd(dy, x, g, mean, rstd, None, None, ad, True)[1]


def timed(fn, rounds=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


n = ROWS * C * 2
out = {"shape": [ROWS, C], "partial_rows": part.shape[1]}
t = timed(lambda: C_.layernorm_fwd(x, g, b, 1e-5)); out["fwd_us"] = t; out["fwd_tbs"] = 2 * n / t / 1e6
t = timed(lambda: C_.layernorm_bwd(dy, x, g, mean, rstd, None, None, None, True)); out["bwd_us"] = t; out["bwd_tbs"] = 3 * n / t / 1e6
t = timed(lambda: C_.layernorm_bwd(dy, x, g, mean, rstd, None, None, ad, True)); out["bwd_addend_us"] = t; out["bwd_addend_tbs"] = 4 * n / t / 1e6
t = timed(lambda: C_.ln_fold_partials(part, dga, dba)); out["fold_us"] = t
print(json.dumps(out))
