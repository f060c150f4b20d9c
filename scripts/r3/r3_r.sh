#!/bin/bash
OUT=gpurun_out/r3r; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
T="tests/test_gpu_e2e.py::test_optimizer_overlapped_with_backward_follows_the_same_trajectory"
for i in 1 2 3; do timeout -k 10 200 python -u -m pytest "$T" -q --timeout 120 --timeout-method thread 2>&1 | grep -E "passed|failed|AssertionError:" | tee -a $OUT/flaky.txt; done
echo "--- group off, wide gemm 700" | tee -a $OUT/flaky.txt
for i in 1 2 3; do DPA_WGRAD_GROUP=0 DPA_WGRAD_WIDE=0 DPA_WIDE_MIN_TILES=700 timeout -k 10 200 python -u -m pytest "$T" -q --timeout 120 --timeout-method thread 2>&1 | grep -E "passed|failed|AssertionError:" | tee -a $OUT/flaky.txt; done
exit 0
