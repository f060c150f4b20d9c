#!/bin/bash
# round-3 GPU pass H: peer-mapped all-reduce, every rank on the one GPU
set -o pipefail
OUT=gpurun_out/r3h; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -u -m pytest tests/test_gpu_peer_allreduce.py -x -v --timeout 150 --timeout-method thread > $OUT/pytest_peer.log 2>&1 || { tail -60 $OUT/pytest_peer.log; exit 1; }
tail -8 $OUT/pytest_peer.log
