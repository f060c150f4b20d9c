#!/bin/bash
# pacer wait: polling with short sleeps (default) vs hipEventSynchronize, same box, every workload
OUT=gpurun_out/r3bx; mkdir -p $OUT
one() { local tag=$1; shift; timeout -k 10 300 python bench.py "$@" > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$tag $*:', d['value'], 'wall ms/step', d['ms_per_step'], 'host cpu ms/step', c['host_cpu_ms_per_step'])" | tee -a $OUT/ab.txt; }
for i in 1 2 3; do
  for w in poll sync; do
    DPA_PACER_WAIT=$w one $w --steps 16 --warmup 4
    DPA_PACER_WAIT=$w one $w --model gpt2 --steps 30 --warmup 5
    DPA_PACER_WAIT=$w one $w --model gpt2 --batch 8 --steps 30 --warmup 5
    DPA_PACER_WAIT=$w one $w --model mnist --steps 2000 --warmup 100
  done
done
