#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3u; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== ${TAG} BN=192 forced" | tee -a $OUT/gemm_epi.txt
DPA_WIDE_BN=192 EPI=1 TOKENS=16384 ROUNDS=3 ONLY=fc timeout -k 10 300 python -u scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_epi.txt || exit 1
