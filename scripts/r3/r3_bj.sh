#!/bin/bash
# the block-count heuristics at the new default batch (1024): staging variant, weight-gradient split target; img/s, one box
OUT=gpurun_out/r3bj; mkdir -p $OUT
one() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 16 --warmup 4 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "$label $(grep -o '"value": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/ab.txt; }
for i in 1 2; do
one default X=1
one igemm_stg=3 DPA_IGEMM_STG=3
one igemm_stg=2 DPA_IGEMM_STG=2
one wgrad_blocks=768 DPA_WGRAD_BLOCKS=768
one wgrad_blocks=1024 DPA_WGRAD_BLOCKS=1024
one tail_blocks=512 DPA_TAIL_BWD_BLOCKS=512
one wgrad_group_tiles=200 DPA_WGRAD_GROUP_TILES=200
done
