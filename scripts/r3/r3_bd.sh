#!/bin/bash
# knob sweep on the ResNet-50 step at batch 512 (defaults were partly chosen at batch 256); img/s, one box
OUT=gpurun_out/r3bd; mkdir -p $OUT
one() {  # label, env assignments...
  local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "$label $(grep -o '"value": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/sweep.txt
}
one default X=1
one wgrad_blocks=384 DPA_WGRAD_BLOCKS=384
one wgrad_blocks=768 DPA_WGRAD_BLOCKS=768
one wgrad_blocks=1024 DPA_WGRAD_BLOCKS=1024
one igemm_stg=2 DPA_IGEMM_STG=2
one igemm_stg=3 DPA_IGEMM_STG=3
one wgrad_dma=1 DPA_WGRAD_DMA=1
one wgrad_dma=0 DPA_WGRAD_DMA=0
one tail_blocks=512 DPA_TAIL_BWD_BLOCKS=512
one bn_rev=3 DPA_BN_REV=3
one bn_rev=0 DPA_BN_REV=0
one wgrad_halo=0 DPA_WGRAD_HALO=0
one conv_halo=0 DPA_CONV_HALO=0
one ds_side=0 DPA_DS_SIDE=0
one wgrad_group_tiles=200 DPA_WGRAD_GROUP_TILES=200
one default X=1
