#!/bin/bash
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3an; mkdir -p $OUT
for rep in 1 2; do for c in 8192 16384; do for B in 32 48; do
DPA_LMHEAD_CHUNK_DEFAULT=$c timeout -k 10 300 python -u bench.py --model gpt2 --batch $B --steps 10 --warmup 3 > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python - <<PY | tee -a $OUT/ab.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("chunk $c batch $B:", d["value"], d["ms_per_step"], "loss", d["final_loss"])
PY
done; done; done
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "lm_head or lmhead or chunk" --timeout 180 --timeout-method thread 2>&1 | tail -3
