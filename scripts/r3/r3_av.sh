#!/bin/bash
# ablation of the (unpipelined) forward attention kernel: which part is it waiting for?  (results are WRONG by construction)
OUT=gpurun_out/r3av; mkdir -p $OUT
for abl in 0 1 2 3 4 5 6 7 0; do
  DPA_ATTN_FWD_PIPE=0 DPA_ATTN_ABL=$abl timeout -k 10 200 python -u scripts/bench_attn.py 16 > $OUT/one.txt 2>&1 || { cat $OUT/one.txt; exit 1; }
  echo "ABL=$abl $(grep attention $OUT/one.txt | cut -c1-80)" | tee -a $OUT/abl.txt
done
