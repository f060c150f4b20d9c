#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3aa; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -u -m pytest tests/test_gpu_gemm_wide.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4 | tee $OUT/pytest.txt || exit 1
for rep in 1 2; do for pw in 0 1; do
echo "== DPA_WIDE_PREW=$pw" | tee -a $OUT/gemm.txt
DPA_WIDE_PREW=$pw TOKENS=16384 ROUNDS=3 ONLY=7 timeout -k 10 300 python -u scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | cut -c1-140 | tee -a $OUT/gemm.txt || exit 1
done; done
for rep in 1 2; do for pw in 0 1; do
DPA_WIDE_PREW=$pw timeout -k 10 300 python -u bench.py --model gpt2 --steps 12 --warmup 4 > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python - <<PY | tee -a $OUT/ab.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("prew $pw:", d["value"], d["ms_per_step"], "loss", d["final_loss"])
PY
done; done
