#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r3ac
timeout -k 10 400 python -u scripts/flaky_overlap.py 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ac/plain.txt | tail -45
BIG=1 timeout -k 10 400 python -u scripts/flaky_overlap.py 25 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ac/big.txt | grep -E "<--|worst"
exit 0
