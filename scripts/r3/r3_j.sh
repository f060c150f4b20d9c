#!/bin/bash
# GPT-2 "alone" kernel times (side streams off: every kernel by itself) + attention micro-benchmark at batch 16
set -o pipefail
OUT=gpurun_out/r3j; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DPA_WGRAD_STREAM=0 DPA_DS_SIDE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o gpt2_alone -- python3 bench.py --model gpt2 --steps 4 --warmup 2 --opt-overlap off > $OUT/alone_prof.log 2>&1 || { tail -20 $OUT/alone_prof.log; exit 1; }
tail -1 $OUT/alone_prof.log | cut -c1-200
timeout -k 10 100 python -u scripts/bench_attn.py 16 1024 12 > $OUT/attn_b16.txt 2>&1 || exit 1
timeout -k 10 100 python -u scripts/bench_attn.py 8 1024 12 >> $OUT/attn_b16.txt 2>&1 || exit 1
cat $OUT/attn_b16.txt
