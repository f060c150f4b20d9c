#!/bin/bash
# ResNet-50 / -152: late-stage 1 x 1 weight gradients through the wide multi-layer kernel, off / on interleaved
set -o pipefail
OUT=gpurun_out/r3ai; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for w in 0 1; do for m in resnet50 resnet152; do
  DPA_WGRAD_WIDE_1X1=$w timeout -k 10 300 python -u bench.py --model $m --steps 12 --warmup 4 > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
  python - <<PY | tee -a $OUT/ab.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("wide 1x1 wgrad $w $m:", d["value"], d["ms_per_step"], "loss", d["final_loss"])
PY
done; done; done
