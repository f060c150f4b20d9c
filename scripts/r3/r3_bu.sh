#!/bin/bash
# unbounded run-ahead, 20 runs: in the slow ones, is the host busy or blocked?  (host CPU ms per step beside the wall time)
OUT=gpurun_out/r3bu; mkdir -p $OUT
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do
  timeout -k 10 300 python bench.py --steps 16 --warmup 4 --run-ahead 0 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], 'wall ms/step', d['ms_per_step'], 'host cpu ms/step', c['host_cpu_ms_per_step'], 'reserved', c['max_memory_reserved_gb'])" | tee -a $OUT/ab.txt
done
timeout -k 10 300 python bench.py --steps 16 --warmup 4 > $OUT/b.txt 2>&1 && grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('paced:', d['value'], 'wall ms/step', d['ms_per_step'], 'host cpu ms/step', c['host_cpu_ms_per_step'])" | tee -a $OUT/ab.txt
