#!/bin/bash
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_gpu_wgrad_wide.py -x -q --timeout 120 --timeout-method thread -k grouped 2>&1 | tail -15
