#!/bin/bash
# paced host with blocking events: speed and host CPU per step, every workload, twice
OUT=gpurun_out/r3bv; mkdir -p $OUT
one() { timeout -k 10 300 python bench.py "$@" > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$*:', d['value'], 'wall ms/step', d['ms_per_step'], 'host cpu ms/step', c['host_cpu_ms_per_step'])" | tee -a $OUT/ab.txt; }
for i in 1 2 3 4 5 6; do one --steps 16 --warmup 4; done
for i in 1 2; do one --batch 512 --steps 20 --warmup 5; one --model gpt2 --steps 20 --warmup 5; one --model gpt2 --batch 8 --steps 20 --warmup 5; one --model resnet152 --steps 8 --warmup 3; one --model mnist --steps 500 --warmup 50; done
