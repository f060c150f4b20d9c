#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r3ak
timeout -k 10 400 python -u -m pytest tests/test_gpu_wgrad_wide.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3ak/pytest.txt 2>&1
tail -12 gpurun_out/r3ak/pytest.txt
