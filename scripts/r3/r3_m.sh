#!/bin/bash
# per-GPU batch scan of GPT-2 / ResNet-152 and the stock comparison line at the larger batches (one box)
set -o pipefail
OUT=gpurun_out/r3m; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # model batch steps
  timeout -k 10 300 python -u bench.py --model $1 --batch $2 --steps $3 --warmup 4 > $OUT/b_$1_$2.log 2>&1 || { echo "$1 batch $2 FAILED"; tail -5 $OUT/b_$1_$2.log; return 1; }
  python - <<PY | tee -a $OUT/batch_scan.txt
import json
d = json.loads([l for l in open("$OUT/b_$1_$2.log") if l.startswith("{")][-1])
print("$1 batch $2:", d["value"], d["unit"], d["ms_per_step"], "ms/step mfu", d.get("mfu"))
PY
}
for b in 16 24 32 48 16 32; do run gpt2 $b 12 || exit 1; done
for b in 256 384 512 256 512; do run resnet152 $b 8 || exit 1; done
timeout -k 10 400 python -u scripts/baseline_torch.py --batch 1024 --steps 8 --warmup 3 2>&1 | tail -3 | tee -a $OUT/stock.txt || exit 1
timeout -k 10 400 python -u scripts/baseline_torch.py --model gpt2 --batch 32 --steps 8 --warmup 3 2>&1 | tail -3 | tee -a $OUT/stock.txt || exit 1
timeout -k 10 400 python -u scripts/baseline_torch.py --model resnet152 --batch 512 --steps 6 --warmup 3 2>&1 | tail -3 | tee -a $OUT/stock.txt || exit 1
