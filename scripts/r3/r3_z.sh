#!/bin/bash
# GPT-2: how many weight-gradient tiles to hold back per launch (96 = one block, split 2 + fold; 200 = two blocks, no split)
set -o pipefail
OUT=gpurun_out/r3z; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do for gt in 96 200; do for B in 16 32; do
  DPA_WGRAD_GROUP_TILES=$gt timeout -k 10 300 python -u bench.py --model gpt2 --steps 12 --warmup 4 --batch $B > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
  python - <<PY | tee -a $OUT/ab.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("group tiles $gt batch $B:", d["value"], d["ms_per_step"], "loss", d["final_loss"])
PY
done; done; done
