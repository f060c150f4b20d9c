#!/bin/bash
# kernel table of the final default (ResNet-50, per-GPU batch 1024) + ResNet-152 at 1024 for the record
export GPU_MAX_HW_QUEUES=8
bash scripts/gpu_profile.sh r50_b1024 || exit 1
mkdir -p gpurun_out/r3bi
timeout -k 10 300 python bench.py --model resnet152 --batch 1024 --steps 8 --warmup 3 > gpurun_out/r3bi/r152_b1024.txt 2>&1 || { tail -5 gpurun_out/r3bi/r152_b1024.txt; exit 1; }
grep -o '"value": [0-9.]*' gpurun_out/r3bi/r152_b1024.txt
