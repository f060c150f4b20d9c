#!/bin/bash
# attention backward: dK/dV kernel at 3 waves/SIMD (168 registers, small scratch) vs 2; branchless diagonal masks
OUT=gpurun_out/r3_ap; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "attention" --timeout 120 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -20 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
DPA_ATTN_BWD_KV_OCC=3 timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -x -q -k "attention" --timeout 120 --timeout-method thread > $OUT/pytest3.txt 2>&1 || { tail -20 $OUT/pytest3.txt; exit 1; }
tail -3 $OUT/pytest3.txt
for occ in 3 2 3 2; do
  for b in 16 8; do
    echo "kv occ=$occ" >> $OUT/attn.txt
    DPA_ATTN_BWD_KV_OCC=$occ timeout -k 10 200 python -u scripts/bench_attn.py $b > $OUT/one.txt 2>&1 || { cat $OUT/one.txt; exit 1; }
    grep attention $OUT/one.txt >> $OUT/attn.txt
  done
done
cat $OUT/attn.txt
for occ in 3 2 3 2; do
  DPA_ATTN_BWD_KV_OCC=$occ timeout -k 10 300 python bench.py --model gpt2 --steps 30 --warmup 10 > $OUT/g.txt 2>&1 || { tail -5 $OUT/g.txt; exit 1; }
  echo "kv occ=$occ $(grep -o '"value": [0-9.]*' $OUT/g.txt)" | tee -a $OUT/gpt2.txt
done
