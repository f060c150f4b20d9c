#!/bin/bash
# host run-ahead bounded to two steps (default now) against unbounded, the default bench, one box
OUT=gpurun_out/r3bo; mkdir -p $OUT
one() { timeout -k 10 300 python bench.py --steps 16 --warmup 4 "$@" > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('run_ahead', c['host_run_ahead_steps'], 'batch', c['per_gpu_batch'], d['value'], d['ms_per_step'], 'reserved', c['max_memory_reserved_gb'], 'device_allocs', c['device_allocs'])" | tee -a $OUT/ab.txt; }
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do one; done
for i in 1 2 3 4 5 6 7 8; do one --run-ahead 0; done
for i in 1 2 3 4; do one --batch 512; done
for i in 1 2; do one --model gpt2; one --model resnet152 --steps 8 --warmup 3; one --model mnist --steps 500 --warmup 50; done
