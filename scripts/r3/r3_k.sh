#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3k; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm_wide.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest_wide.log 2>&1 || { tail -40 $OUT/pytest_wide.log; exit 1; }
tail -2 $OUT/pytest_wide.log
for tw in 0 1 0 1; do
  for b in 16 8; do
    DPA_GELU_TWIN=$tw timeout -k 10 200 python -u bench.py --model gpt2 --batch $b --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('twin $tw batch $b:', d['value'], d['ms_per_step'])" | tee -a $OUT/gelu_twin_ab.txt || exit 1
  done
done
