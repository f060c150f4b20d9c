#!/bin/bash
# focused repeat of the two candidates of r3_bd.sh
OUT=gpurun_out/r3be; mkdir -p $OUT
one() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "$label $(grep -o '"value": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/ab.txt; }
for i in 1 2 3 4; do
one default X=1
one bn_rev=0 DPA_BN_REV=0
one bn_rev=3 DPA_BN_REV=3
one wgrad_halo=0 DPA_WGRAD_HALO=0
done
