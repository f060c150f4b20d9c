#!/bin/bash
# is there a sporadic slow mode at batch 1024?  default config 12 times, forced STG=2 8 times, batch 512 6 times
OUT=gpurun_out/r3bk; mkdir -p $OUT
one() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 16 --warmup 4 "${EXTRA[@]}" > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "$label $(grep -o '"value": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/ab.txt; }
EXTRA=()
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do one default X=1; done
for i in 1 2 3 4 5 6 7 8; do one igemm_stg=2 DPA_IGEMM_STG=2; done
EXTRA=(--batch 512)
for i in 1 2 3 4 5 6; do one b512 X=1; done
