#!/bin/bash
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3am; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_wgrad_wide.py -x -q --timeout 180 --timeout-method thread > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
for h in 0 1 0 1; do
echo "== hybrid $h" | tee -a $OUT/bench.txt
DPA_WGRAD_WIDE_HYBRID=$h LMHEAD=1 ROUNDS=3 timeout -k 10 200 python -u scripts/bench_wgrad_wide.py 2>&1 | grep "one launch " | tee -a $OUT/bench.txt
done
for rep in 1 2; do for h in 0 1; do
DPA_WGRAD_WIDE_HYBRID=$h timeout -k 10 300 python -u bench.py --model gpt2 --steps 12 --warmup 4 > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python - <<PY | tee -a $OUT/ab.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("hybrid $h:", d["value"], d["ms_per_step"], "loss", d["final_loss"])
PY
done; done
