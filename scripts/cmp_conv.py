#!/usr/bin/env python
"""Side-by-side of two bench_conv.py JSON outputs (us per layer and per-step totals)."""
import json
import sys

a, b = (json.load(open(p)) for p in sys.argv[1:3])
for ra, rb in zip(a["rows"], b["rows"]):
    print(f"{ra['name']:24s} x{ra['count']} | f {ra['fprop_us']:7.1f} -> {rb['fprop_us']:7.1f} | d {ra['dgrad_us']:7.1f} -> {rb['dgrad_us']:7.1f}"
          f" | w {ra['wgrad_us']:7.1f} -> {rb['wgrad_us']:7.1f}")
print({k: round(v, 2) for k, v in a["totals_ms"].items()}, "->", {k: round(v, 2) for k, v in b["totals_ms"].items()})
