#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs written by pmc_conv.sh: per-kernel mean of every counter."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "conv_"
vals = defaultdict(list)
dur = []
for f in sorted(glob.glob(d + "/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(d + "/p*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(d, "kernel us (profiled):", [round(x, 1) for x in dur[:6]])
m = {k: sum(v) / len(v) for k, v in vals.items()}
for k in sorted(m):
    print(f"  {k:28s} {m[k]:16.0f}")
if "SQ_WAVES" in m:
    w = m["SQ_WAVES"]
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_MFMA"):
        if k in m:
            print(f"  per wave {k:20s} {m[k] / w:10.1f}")
    if "SQ_WAVE_CYCLES" in m:
        print(f"  wave-cycles per wave (quad-cycles x4): {m['SQ_WAVE_CYCLES'] * 4 / w:10.0f}")
