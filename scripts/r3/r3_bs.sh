#!/bin/bash
# stock PyTorch-ROCm GPT-2 small at 32 x 1024, and this framework at 32 / 16 beside it (three runs each)
OUT=gpurun_out/r3bs; mkdir -p $OUT
timeout -k 10 600 python -u scripts/baseline_torch.py --model gpt2 --steps 10 --warmup 3 --batch 32 > $OUT/stock_gpt2_b32.txt 2>&1 || { tail -5 $OUT/stock_gpt2_b32.txt; exit 1; }
tail -1 $OUT/stock_gpt2_b32.txt
for i in 1 2 3; do
  for b in 32 16; do
    timeout -k 10 300 python bench.py --model gpt2 --batch $b --steps 16 --warmup 4 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
    grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('batch', c['per_gpu_batch'], d['value'], d['ms_per_step'], 'ms mfu', d['mfu'], 'allocated', c['max_memory_allocated_gb'])" | tee -a $OUT/ab.txt
  done
done
