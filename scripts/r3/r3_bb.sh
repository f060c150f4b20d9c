#!/bin/bash
OUT=gpurun_out/r3bb; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_gpu_e2e.py -q -x --timeout 180 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
