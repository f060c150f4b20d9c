#!/bin/bash
OUT=gpurun_out/r3aw; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_real_shapes.py -x -q -k "attention or attn or gpt2" --timeout 120 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for cfg in "16 1024 12" "8 1024 12"; do
  timeout -k 10 200 python -u scripts/bench_attn.py $cfg > $OUT/one.txt 2>&1 || { cat $OUT/one.txt; exit 1; }
  grep attention $OUT/one.txt | tee -a $OUT/attn.txt
done
timeout -k 10 300 python bench.py --model gpt2 --steps 30 --warmup 10 > $OUT/g.txt 2>&1 || { tail -5 $OUT/g.txt; exit 1; }
grep -o '"value": [0-9.]*' $OUT/g.txt
