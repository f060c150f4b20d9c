#!/usr/bin/env python
"""A/B of the halo-tiled stage-1 3 x 3 convolution (csrc/conv3x3_halo.hip) against the implicit GEMM it replaces, in ONE
process with interleaved rounds (random operands; BatchNorm-statistics epilogues on, as in the training step)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.ops import _ext  # noqa: E402


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[512, 256])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    C = _ext.load()
    rows = []
    for N in args.batch:
        x = (torch.randn(N, 56, 56, 64, device="cuda")).to(torch.bfloat16)
        dy = (torch.randn(N, 56, 56, 64, device="cuda")).to(torch.bfloat16)
        cx = (torch.randn(N, 56, 56, 64, device="cuda")).to(torch.bfloat16)
        w = (torch.randn(64, 3, 3, 64, device="cuda") * 0.05).to(torch.bfloat16)
        wt = C.transpose_filter(w)
        mask = torch.randint(0, 256, (x.numel() // 8,), device="cuda", dtype=torch.uint8)
        sb = torch.zeros(32, 2, 64, device="cuda")
        flop = 2.0 * N * 56 * 56 * 64 * 64 * 9
        ops = {
            "fprop+stats": lambda: C.conv_fprop(x, w, None, 1, 1, 0, True, sb),
            "dgrad+bn-bwd stats": lambda: C.conv_dgrad(dy, wt, 56, 56, 1, 1, None, None, cx, mask, sb),
            "dgrad plain": lambda: C.conv_dgrad(dy, wt, 56, 56, 1, 1, None, None),
        }
        for name, fn in ops.items():
            t = {True: [], False: []}
            for on in (True, False):
                C.set_conv3x3_halo(on); fn(); fn()
            torch.cuda.synchronize()
            for _ in range(args.rounds):
                for on in (True, False):
                    C.set_conv3x3_halo(on)
                    t[on].append(timed(fn, args.iters))
            C.set_conv3x3_halo(True)
            med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
            row = {"batch": N, "op": name, "halo_us": round(med[True], 1), "igemm_us": round(med[False], 1),
                   "halo_tflops": round(flop / med[True] / 1e6, 1), "igemm_tflops": round(flop / med[False] / 1e6, 1),
                   "halo_min_us": round(min(t[True]), 1), "igemm_min_us": round(min(t[False]), 1)}
            rows.append(row)
            print(f"batch {N:4d} {name:20s}: halo {row['halo_us']:7.1f} us ({row['halo_tflops']:6.1f} TFLOP/s)   "
                  f"implicit GEMM {row['igemm_us']:7.1f} us ({row['igemm_tflops']:6.1f} TFLOP/s)   x{med[False] / med[True]:.2f}", flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
