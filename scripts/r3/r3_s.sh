#!/bin/bash
OUT=gpurun_out/r3s; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread > $OUT/pytest.txt 2>&1; rc=$?
tail -12 $OUT/pytest.txt
exit $rc
