#!/bin/bash
# implicit-GEMM epilogue: non-temporal output stores above a size threshold (MB), ResNet-50 / ResNet-152 step
OUT=gpurun_out/r3ax; mkdir -p $OUT
for rep in 1 2; do
for mb in 0 600 300 150; do
  DPA_IGEMM_NTS_MB=$mb timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "r50 nts_mb=$mb $(grep -o '"value": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/ab.txt
done
done
for mb in 0 300; do
  DPA_IGEMM_NTS_MB=$mb timeout -k 10 300 python bench.py --model resnet152 --steps 10 --warmup 3 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "r152 nts_mb=$mb $(grep -o '"value": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/ab.txt
done
