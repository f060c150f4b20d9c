#!/bin/bash
# ResNet-152 with its new default batch (1024), three runs, 512 beside it
OUT=gpurun_out/r3br; mkdir -p $OUT
for i in 1 2 3; do
  for b in "" "--batch 512"; do
    timeout -k 10 300 python bench.py --model resnet152 --steps 8 --warmup 3 $b > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
    grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('batch', c['per_gpu_batch'], d['value'], d['ms_per_step'], 'ms vs_baseline', d['vs_baseline'], 'mfu', d['mfu'], 'allocated', c['max_memory_allocated_gb'], 'reserved', c['max_memory_reserved_gb'])" | tee -a $OUT/ab.txt
  done
done
