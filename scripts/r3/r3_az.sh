#!/bin/bash
# slowest GPU tests on a cold box
OUT=gpurun_out/r3az; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 180 --timeout-method thread --durations=15 > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
grep -A 18 "slowest" $OUT/pytest.txt; tail -1 $OUT/pytest.txt
