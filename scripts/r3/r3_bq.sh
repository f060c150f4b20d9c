#!/bin/bash
# stock PyTorch-ROCm ResNet-152 at per-GPU batch 1024
OUT=gpurun_out/r3bq; mkdir -p $OUT
timeout -k 10 1120 python -u scripts/baseline_torch.py --model resnet152 --steps 6 --warmup 2 --batch 1024 > $OUT/stock_r152_b1024.txt 2>&1
rc=$?
tail -3 $OUT/stock_r152_b1024.txt
exit $rc
