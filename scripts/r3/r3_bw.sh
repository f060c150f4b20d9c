#!/bin/bash
# pacer events: spinning (default) vs blocking flag, same box, GPT-2 and ResNet-50
OUT=gpurun_out/r3bw; mkdir -p $OUT
one() { local tag=$1; shift; timeout -k 10 300 python bench.py "$@" > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$tag $*:', d['value'], 'wall ms/step', d['ms_per_step'], 'host cpu ms/step', c['host_cpu_ms_per_step'])" | tee -a $OUT/ab.txt; }
for i in 1 2 3; do
  DPA_PACER_BLOCKING=0 one spin --model gpt2 --steps 30 --warmup 5
  DPA_PACER_BLOCKING=1 one block --model gpt2 --steps 30 --warmup 5
  DPA_PACER_BLOCKING=0 one spin --model gpt2 --steps 30 --warmup 5 --run-ahead 0
  DPA_PACER_BLOCKING=0 one spin --steps 16 --warmup 4
  DPA_PACER_BLOCKING=1 one block --steps 16 --warmup 4
done
