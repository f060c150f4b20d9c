#!/bin/bash
# attention forward: software-pipelined half-tile kernel (PIPE=1) vs the unpipelined one (PIPE=0)
OUT=gpurun_out/r3au; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_real_shapes.py -x -q -k "attention or attn or gpt2" --timeout 120 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for pp in 1 0 1 0; do
  for cfg in "16 1024 12" "8 1024 12" "4 4096 12"; do
    DPA_ATTN_FWD_PIPE=$pp timeout -k 10 200 python -u scripts/bench_attn.py $cfg > $OUT/one.txt 2>&1 || { cat $OUT/one.txt; exit 1; }
    echo "PIPE=$pp $(grep attention $OUT/one.txt)" | tee -a $OUT/attn.txt
  done
done
for pp in 1 0 1 0; do
  DPA_ATTN_FWD_PIPE=$pp timeout -k 10 300 python bench.py --model gpt2 --steps 30 --warmup 10 > $OUT/g.txt 2>&1 || { tail -5 $OUT/g.txt; exit 1; }
  echo "PIPE=$pp $(grep -o '"value": [0-9.]*' $OUT/g.txt)" | tee -a $OUT/gpt2.txt
done
