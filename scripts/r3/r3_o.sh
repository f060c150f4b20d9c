#!/bin/bash
# GPT-2 step with the mid-size GEMMs on the wide kernel (DPA_WIDE_MIN_TILES lowered) vs default, interleaved
set -o pipefail
OUT=gpurun_out/r3o; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
for mt in 700 256 192; do
for b in 16 32; do
  DPA_WIDE_MIN_TILES=$mt timeout -k 10 300 python -u bench.py --model gpt2 --batch $b --steps 12 --warmup 4 > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
  python - <<PY | tee -a $OUT/ab.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("min_tiles $mt batch $b:", d["value"], d["ms_per_step"], "loss", d["final_loss"])
PY
done; done; done
