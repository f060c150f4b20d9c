#!/bin/bash
# same box: bench.py (batch 512, then the default 1024) ten times each, the in-process probe in between
OUT=gpurun_out/r3bm; mkdir -p $OUT
hostname | tee -a $OUT/ab.txt
one() { local label=$1; shift
  timeout -k 10 300 python bench.py --steps 16 --warmup 4 "$@" > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "$label $(grep -o '"value": [0-9.]*' $OUT/b.txt) $(grep -o '"ms_per_step": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/ab.txt; }
for i in 1 2 3 4 5 6 7 8 9 10; do one b512 --batch 512; done
for i in 1 2 3; do timeout -k 10 200 python -u scripts/slowmode_probe.py 512 40 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt; done
for i in 1 2 3 4 5 6 7 8 9 10; do one default; done
for i in 1 2 3; do timeout -k 10 200 python -u scripts/slowmode_probe.py 1024 30 2>&1 | grep -v amdgpu | tee -a $OUT/ab.txt; done
