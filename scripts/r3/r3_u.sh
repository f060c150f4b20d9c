#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3u; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -u -m pytest tests/test_gpu_gemm_wide.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4 | tee $OUT/pytest.txt || exit 1
echo "== ${TAG}" | tee -a $OUT/gemm_epi.txt
EPI=1 TOKENS=16384 ROUNDS=3 ONLY=${ONLY:-7} timeout -k 10 300 python -u scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | grep -v lm_head | tee -a $OUT/gemm_epi.txt || exit 1
for rep in 1 2; do for tw in 1 0; do
DPA_GELU_TWIN=$tw timeout -k 10 300 python -u bench.py --model gpt2 --steps 12 --warmup 4 > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python - <<PY | tee -a $OUT/ab.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("${TAG} twin $tw:", d["value"], d["ms_per_step"], "loss", d["final_loss"])
PY
done; done
