#!/usr/bin/env python3
"""GPU busy fraction of the last training step in a rocprofv3 kernel trace: union of kernel intervals / wall time
between two optimizer kernels, plus the largest idle gaps and what follows them.

    python scripts/busy_fraction.py gpurun_out/prof/r152_kernel_trace.csv [marker-kernel-substring]
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
busy, cur_end, gaps = 0, t0, []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        gaps.append((s - cur_end, re.sub(r"\(.*", "", r["Kernel_Name"])[-60:]))
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
print(f"kernels {len(step)}  wall {(t1 - t0) / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  ({100.0 * busy / (t1 - t0):.1f} %)  "
      f"kernel-time sum {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6:.3f} ms")
gaps.sort(reverse=True)
print("idle total %.3f ms in %d gaps; largest:" % (sum(g for g, _ in gaps) / 1e6, len(gaps)))
for g, n in gaps[:12]:
    print(f"  {g / 1e3:8.1f} us before {n}")
