#!/bin/bash
# attention kernels at longer sequences
OUT=gpurun_out/r3as; mkdir -p $OUT
for cfg in "4 4096 12" "2 8192 12" "1 16384 12" "16 1024 12"; do
  timeout -k 10 200 python -u scripts/bench_attn.py $cfg > $OUT/one.txt 2>&1 || { cat $OUT/one.txt; exit 1; }
  grep attention $OUT/one.txt | tee -a $OUT/attn_long.txt
done
