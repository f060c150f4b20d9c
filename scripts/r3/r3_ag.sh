#!/bin/bash
# GPT-2 small as a 2-stage pipeline, both stages on the one GPU over gloo: the wide weight gradients + batcher inside the pipeline engine
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r3ag
DPA_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 2 --backend gloo --model gpt2 --pipeline 2 --micro-batches 2 --batch 4 --steps 3 --warmup 1 > gpurun_out/r3ag/pp2_gpt2.log 2>&1
echo "rc=$?"; grep '^{' gpurun_out/r3ag/pp2_gpt2.log | tail -1 | cut -c1-400; tail -5 gpurun_out/r3ag/pp2_gpt2.log | cut -c1-300
python - <<'PY'
import torch
from distributedpytorch_amd.ops import wgroup
print("ok import")
PY
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
