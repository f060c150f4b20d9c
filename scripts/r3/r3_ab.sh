#!/bin/bash
# PMC counters of the wide GEMM / wide weight gradient main loops (long-K plain shapes)
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash scripts/pmc_run.sh wide_fc2 "gemm_wide" scripts/one_gemm.py 16384 3072 768 wide fprop 10 || exit 1
bash scripts/pmc_run.sh wide_lmdgrad "gemm_wide" scripts/one_gemm.py 16384 50304 768 wide fprop 3 || exit 1
