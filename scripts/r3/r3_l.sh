#!/bin/bash
# per-GPU batch scan of the ResNet-50 step (one box), memory included
set -o pipefail
OUT=gpurun_out/r3l; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 512 768 1024 512 768 1024; do
  timeout -k 10 300 python -u - <<PY | tee -a $OUT/batch_scan.txt || exit 1
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--batch", "$b", "--steps", "12", "--warmup", "4"], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("batch $b failed:", out.stderr[-400:]); sys.exit(1)
d = json.loads(line[-1])
print("batch $b:", d["value"], "img/s", d["ms_per_step"], "ms/step mfu", d.get("mfu"))
PY
done
