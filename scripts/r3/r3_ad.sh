#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
T="tests/test_gpu_e2e.py::test_optimizer_overlapped_with_backward_follows_the_same_trajectory"
for i in 1 2 3 4 5 6 7 8; do timeout -k 10 200 python -u -m pytest "$T" -q --timeout 120 --timeout-method thread 2>&1 | grep -E "passed|failed|Error" ; done
exit 0
