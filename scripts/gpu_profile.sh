#!/bin/bash
# rocprofv3 kernel-trace stats of a bench workload (default: ResNet-50, 1 GPU).
#   scripts/gpu_profile.sh [tag] [bench.py args...]
set -o pipefail
TAG=${1:-r50}; shift
mkdir -p gpurun_out/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --steps 4 --warmup 2 "$@" > gpurun_out/prof/${TAG}_under_prof.log 2>&1 || { tail -20 gpurun_out/prof/${TAG}_under_prof.log; exit 1; }
tail -2 gpurun_out/prof/${TAG}_under_prof.log
find gpurun_out/prof -name "${TAG}*stats*" | head
