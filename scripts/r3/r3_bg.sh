#!/bin/bash
# stock PyTorch-ROCm ResNet-50 at per-GPU batch 1024 (MIOpen's search did not finish inside a 10-minute call earlier)
OUT=gpurun_out/r3bg; mkdir -p $OUT
timeout -k 10 1120 python -u scripts/baseline_torch.py --steps 10 --warmup 3 --batch 1024 > $OUT/stock_r50_b1024.txt 2>&1
rc=$?
tail -3 $OUT/stock_r50_b1024.txt
exit $rc
