#!/bin/bash
# One lease on an 8-GPU node answers SURVEY.md section 5.8(a): ResNet-50 DP throughput at 1 / 2 / 4 / 8 GPUs for
# every bucket size x all-reduce algorithm x collective layer, one table at the end.
#
#   scripts/scale_sweep.sh [model] [steps] [warmup]        (default: resnet50 30 8)
#
# Arms: c10d ring (torch.distributed's RCCL process group) | native ring (csrc/rccl_comm.cpp, ncclAllReduce) |
#       native direct fp32 (one-shot reduce-scatter + all-gather over the 7 xGMI links) | native direct bf16 wire.
# Every run prints bench.py's JSON line (with --breakdown: forward / backward / exposed communication / optimizer);
# they are collected in gpurun_out/scale_sweep.jsonl and summarised by scripts/scale_table.py.
set -o pipefail
MODEL=${1:-resnet50}; STEPS=${2:-30}; WARMUP=${3:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/scale_sweep.jsonl
mkdir -p "$ROOT/gpurun_out"; : > "$OUT"
NGPU=$(python3 -c 'import torch; print(torch.cuda.device_count())')
PORT=29610
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # n bucket_mb comm allreduce wire
    local n=$1 mb=$2 comm=$3 ar=$4 wire=$5
    PORT=$((PORT + 1))
    local args="--gpus $n --steps $STEPS --warmup $WARMUP --model $MODEL --bucket-mb $mb --comm $comm --allreduce $ar --grad-wire $wire --breakdown"
    local line
    if [ "$n" = 1 ]; then
        line=$(timeout -k 10 600 python3 "$ROOT/bench.py" $args 2>>"$ROOT/gpurun_out/scale_sweep.err" | grep '^{' | tail -1)
    else
        line=$(timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 \
               --master-port $PORT "$ROOT/bench.py" $args 2>>"$ROOT/gpurun_out/scale_sweep.err" | grep '^{' | tail -1)
    fi
    if [ -z "$line" ]; then line="{\"error\": \"no result\"}"; fi
    echo "{\"n\": $n, \"bucket_mb\": $mb, \"comm\": \"$comm\", \"allreduce\": \"$ar\", \"wire\": \"$wire\", \"result\": $line}" | tee -a "$OUT"
}
for n in 1 2 4 8; do
    [ "$n" -le "$NGPU" ] || { echo "[scale_sweep] $NGPU GPUs here: skipping N=$n" >&2; continue; }
    if [ "$n" = 1 ]; then run 1 32 c10d ring fp32; continue; fi
    for mb in 8 16 32 64; do
        run $n $mb c10d ring fp32
        run $n $mb native ring fp32
        run $n $mb native direct fp32
        run $n $mb native direct bf16
    done
done
python3 "$ROOT/scripts/scale_table.py" "$OUT"
