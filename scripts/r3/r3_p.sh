#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3p; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -u -m pytest tests/test_gpu_wgrad_wide.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -25 | tee $OUT/pytest.txt || exit 1
SPLITS=1,2,3,7 timeout -k 10 200 python -u scripts/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids | tee $OUT/bench.txt || exit 1
TOKENS=32768 SPLITS=2,7 timeout -k 10 200 python -u scripts/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench.txt || exit 1
LMHEAD=1 SPLITS=1,3 timeout -k 10 200 python -u scripts/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bench.txt || exit 1
