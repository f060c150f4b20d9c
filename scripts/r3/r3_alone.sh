#!/bin/bash
# "Alone" kernel times of the ResNet-50 step: side streams off, so every kernel runs by itself and the rocprofv3
# durations add up to the step; plus the per-layer conv table at batch 512.
set -o pipefail
OUT=gpurun_out/r3b; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DPA_WGRAD_STREAM=0 DPA_DS_SIDE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o alone -- python3 bench.py --steps 4 --warmup 2 > $OUT/alone_prof.log 2>&1 || { tail -20 $OUT/alone_prof.log; exit 1; }
timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 > $OUT/bench_noside.log 2>&1 || exit 1
unset DPA_WGRAD_STREAM DPA_DS_SIDE
timeout -k 10 400 python -u scripts/bench_conv.py --batch 512 --iters 10 --out $OUT/bench_conv_b512.json > $OUT/bench_conv_b512.log 2>&1 || { tail -20 $OUT/bench_conv_b512.log; exit 1; }
find $OUT/prof -name "*kernel_stats*" | head
tail -1 $OUT/bench_noside.log
