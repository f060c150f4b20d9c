#!/bin/bash
# attention forward: VALU diet + occupancy 4 vs 2 (tests, kernel bench at B=16 / 8, GPT-2 step A/B)
OUT=gpurun_out/r3_ao; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "attention" --timeout 120 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -20 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
for occ in 4 2 4 2; do
  for b in 16 8; do
    echo "occ=$occ" >> $OUT/attn.txt
    DPA_ATTN_FWD_OCC=$occ timeout -k 10 200 python -u scripts/bench_attn.py $b > $OUT/one.txt 2>&1 || { cat $OUT/one.txt; exit 1; }
    grep attention $OUT/one.txt >> $OUT/attn.txt
  done
done
cat $OUT/attn.txt
for occ in 4 2 4 2; do
  DPA_ATTN_FWD_OCC=$occ timeout -k 10 300 python bench.py --model gpt2 --steps 30 --warmup 10 > $OUT/g.txt 2>&1 || { tail -5 $OUT/g.txt; exit 1; }
  echo "occ=$occ $(grep -o '"value": [0-9.]*' $OUT/g.txt)" | tee -a $OUT/gpt2.txt
done
