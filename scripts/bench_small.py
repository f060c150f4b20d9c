"""Per-kernel times of the small-CNN path (csrc/smallnet.hip) at the reference's batch (128): each launch alone, median
of 200, so a change to one kernel can be judged without the rest of the step.    python scripts/bench_small.py [B]"""
import json
import statistics
import sys

import torch

from distributedpytorch_amd.ops import _ext

C = _ext.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"
x = torch.randn(B, 1, 28, 28, device=dev)
w1, b1 = torch.randn(10, 1, 5, 5, device=dev) * 0.2, torch.randn(10, device=dev)
w2, b2 = torch.randn(20, 10, 5, 5, device=dev) * 0.1, torch.randn(20, device=dev)
f1, fb1 = torch.randn(50, 320, device=dev) * 0.05, torch.randn(50, device=dev)
f2, fb2 = torch.randn(10, 50, device=dev) * 0.1, torch.randn(10, device=dev)
t = torch.randint(0, 10, (B,), device=dev)
a1, i1 = C.conv5_pool_relu_fwd(x, w1, b1, 0.0, 0, 0, None)
a2, i2 = C.conv5_pool_relu_fwd(a1, w2, b2, 0.5, 1, 2, None)
h0 = a2.reshape(B, 320)
h1 = C.linear_f32_fwd(h0, f1, fb1, True, 0.5, 1, 3, None)
lp = C.linear_log_softmax_fwd(h1, f2, fb2)
dlp = C.nll_bwd(torch.ones(1, device=dev), t, B, 10)
dh1 = torch.randn_like(h1); dh0 = torch.randn_like(h0); da1 = torch.randn_like(a1)
gw1, gb1, gw2, gb2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros_like(b2)
gf1, gfb1, gf2, gfb2 = torch.zeros_like(f1), torch.zeros_like(fb1), torch.zeros_like(f2), torch.zeros_like(fb2)
cases = {
    "conv1 fwd": lambda: C.conv5_pool_relu_fwd(x, w1, b1, 0.0, 0, 0, None),
    "conv2 fwd (+dropout2d)": lambda: C.conv5_pool_relu_fwd(a1, w2, b2, 0.5, 1, 2, None),
    "fc1 fwd (+relu+dropout)": lambda: C.linear_f32_fwd(h0, f1, fb1, True, 0.5, 1, 3, None),
    "fc2 + log_softmax fwd": lambda: C.linear_log_softmax_fwd(h1, f2, fb2),
    "nll fwd": lambda: C.nll_fwd(lp, t),
    "nll bwd": lambda: C.nll_bwd(torch.ones(1, device=dev), t, B, 10),
    "fc2+lsm bwd (dx|dw)": lambda: C.linear_f32_bwd(h1, f2, lp, dlp, False, 1.0, True, True, gf2, gfb2),
    "fc1 bwd (dx|dw)": lambda: C.linear_f32_bwd(h0, f1, h1, dh1, True, 2.0, False, True, gf1, gfb1),
    "conv2 bwd (dx|dw)": lambda: C.conv5_pool_relu_bwd(a1, w2, dh0.view_as(a2), a2, i2, 2.0, True, gw2, gb2),
    "conv1 bwd (dw)": lambda: C.conv5_pool_relu_bwd(x, w1, da1, a1, i1, 1.0, False, gw1, gb1),
}
out = {}
for name, fn in cases.items():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    out[name] = round(statistics.median(ts), 2)
out["sum_us"] = round(sum(out.values()), 1)
print(json.dumps(out, indent=1))
