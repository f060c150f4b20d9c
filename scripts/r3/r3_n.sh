#!/bin/bash
# 256 x 192 tile of gemm_wide: tests, then the GEMM table at 16,384 tokens with each tile width forced
set -o pipefail
OUT=gpurun_out/r3n; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -u -m pytest tests/test_gpu_gemm_wide.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -15 | tee $OUT/pytest.txt || exit 1
for bn in 256 192 0; do
  echo "== DPA_WIDE_BN=$bn" | tee -a $OUT/gemm.txt
  DPA_WIDE_BN=$bn TOKENS=16384 ROUNDS=3 ONLY=${ONLY:-7} WITH_LIB=1 timeout -k 10 300 python -u scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm.txt || exit 1
done
