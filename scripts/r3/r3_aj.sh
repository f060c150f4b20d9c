#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r3aj
timeout -k 10 800 python -u -m pytest tests/test_gpu_real_shapes.py tests/test_gpu_e2e.py tests/test_gpu_deterministic.py tests/test_gpu_tail_bwd.py -x -v --timeout 180 --timeout-method thread > gpurun_out/r3aj/pytest.txt 2>&1
tail -5 gpurun_out/r3aj/pytest.txt
