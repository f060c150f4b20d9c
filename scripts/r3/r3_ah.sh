#!/bin/bash
# final GEMM table: wide kernel (tile width and split chosen by plan_wide) vs the 128 x 128 kernel vs hipBLASLt, 16,384 and 8,192 tokens
set -o pipefail
OUT=gpurun_out/r3ah; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for tok in 16384 8192; do
  TOKENS=$tok ROUNDS=5 WITH_LIB=1 OUT=$OUT/gemm_$tok.json timeout -k 10 400 python -u scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_$tok.txt || exit 1
done
