#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of one .hip file, from hipcc's resource-usage remarks.

    python scripts/kernel_regs.py distributedpytorch_amd/csrc/conv_igemm.hip [name-filter]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)\s*(?:\[bytes/lane\]|\[waves/SIMD\]|\[bytes/workgroup\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>5} {'scratch':>7} {'occ':>3} {'LDS':>6}  kernel")
for (k, r), n in zip(rows.items(), names):
    if flt and flt not in n:
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
    print(f"{r.get('VGPRs', 0):>5} {r.get('AGPRs', 0):>5} {r.get('VGPR Spill', 0):>5} {r.get('ScratchSize', 0):>7} "
          f"{r.get('Occupancy', 0):>3} {r.get('LDS Size', 0):>6}  {n}")
