#!/usr/bin/env python
"""Table of scripts/scale_sweep.sh results: throughput, weak-scaling efficiency against the N=1 run of the same
sweep, exposed communication per step."""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = next((r["result"].get("value") for r in rows if r["n"] == 1 and "value" in r["result"]), None)
print(f"{'N':>2s} {'bucket':>6s} {'layer':>7s} {'algo':>7s} {'wire':>5s} | {'items/s':>10s} {'ms/step':>8s} {'eff':>6s} | "
      f"{'fwd':>6s} {'bwd':>6s} {'comm':>6s} {'opt':>6s} | {'MB/step':>8s}")
for r in rows:
    res = r["result"]
    if "value" not in res:
        print(f"{r['n']:2d} {r['bucket_mb']:6} {r['comm']:>7s} {r['allreduce']:>7s} {r['wire']:>5s} | failed")
        continue
    b = res.get("breakdown_ms") or {}
    eff = res["value"] / (r["n"] * base) if base else float("nan")
    print(f"{r['n']:2d} {r['bucket_mb']:6} {r['comm']:>7s} {r['allreduce']:>7s} {r['wire']:>5s} | {res['value']:10.1f} "
          f"{res['ms_per_step']:8.2f} {eff:6.3f} | {b.get('forward', 0):6.2f} {b.get('backward', 0):6.2f} "
          f"{b.get('exposed_comm', 0):6.2f} {b.get('optimizer', 0):6.2f} | {res.get('allreduce_bytes_per_step', 0) / 2 ** 20:8.1f}")
