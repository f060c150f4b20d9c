#!/bin/bash
# stock PyTorch-ROCm comparison line at the larger per-GPU batches (output unpiped: the MIOpen search is silent for minutes)
OUT=gpurun_out/r3y; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u scripts/baseline_torch.py --batch 1024 --steps 8 --warmup 3 > $OUT/stock_r50_1024.txt 2>&1; tail -2 $OUT/stock_r50_1024.txt
timeout -k 10 500 python -u scripts/baseline_torch.py --model resnet152 --batch 512 --steps 6 --warmup 3 > $OUT/stock_r152_512.txt 2>&1; tail -2 $OUT/stock_r152_512.txt
exit 0
