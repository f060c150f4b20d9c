#!/bin/bash
# attention forward: softmax denominator through the MFMA pipe (LM) vs vector additions
OUT=gpurun_out/r3at; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_real_shapes.py -x -q -k "attention or attn or gpt2" --timeout 120 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -20 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for lm in 1 0 1 0; do
  for cfg in "16 1024 12" "8 1024 12" "4 4096 12"; do
    DPA_ATTN_FWD_LM=$lm timeout -k 10 200 python -u scripts/bench_attn.py $cfg > $OUT/one.txt 2>&1 || { cat $OUT/one.txt; exit 1; }
    echo "LM=$lm $(grep attention $OUT/one.txt)" | tee -a $OUT/attn.txt
  done
done
for lm in 1 0 1 0; do
  DPA_ATTN_FWD_LM=$lm timeout -k 10 300 python bench.py --model gpt2 --steps 30 --warmup 10 > $OUT/g.txt 2>&1 || { tail -5 $OUT/g.txt; exit 1; }
  echo "LM=$lm $(grep -o '"value": [0-9.]*' $OUT/g.txt)" | tee -a $OUT/gpt2.txt
done
