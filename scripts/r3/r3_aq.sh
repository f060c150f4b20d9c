#!/bin/bash
# attention: packed fp32 math (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) in the softmax of the forward and the dS of both backward kernels;
# native self-test in phases (comm.py) under the real RCCL path at world size 1
OUT=gpurun_out/r3_aq; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_real_shapes.py -x -q -k "attention or attn or gpt2" --timeout 120 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -20 $OUT/pytest.txt; exit 1; }
tail -3 $OUT/pytest.txt
for i in 1 2; do
  for b in 16 8; do
    timeout -k 10 200 python -u scripts/bench_attn.py $b > $OUT/one.txt 2>&1 || { cat $OUT/one.txt; exit 1; }
    grep attention $OUT/one.txt >> $OUT/attn.txt
  done
done
cat $OUT/attn.txt
for i in 1 2; do
  timeout -k 10 300 python bench.py --model gpt2 --steps 30 --warmup 10 > $OUT/g.txt 2>&1 || { tail -5 $OUT/g.txt; exit 1; }
  echo "$(grep -o '"value": [0-9.]*' $OUT/g.txt)" | tee -a $OUT/gpt2.txt
done
timeout -k 10 300 python -u -m pytest tests/test_gpu_e2e.py -x -q -k "native or rccl or comm or fallback" --timeout 120 --timeout-method thread > $OUT/pytest_comm.txt 2>&1 || { tail -20 $OUT/pytest_comm.txt; exit 1; }
tail -3 $OUT/pytest_comm.txt
