#!/bin/bash
# rocprofv3 kernel-trace stats of the flagship bench (ResNet-50, 1 GPU).
set -o pipefail
mkdir -p gpurun_out/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r50 -- python3 bench.py --steps 4 --warmup 2 > gpurun_out/prof/bench_under_prof.log 2>&1 || { tail -20 gpurun_out/prof/bench_under_prof.log; exit 1; }
tail -2 gpurun_out/prof/bench_under_prof.log
find gpurun_out/prof -name "*stats*" | head
