#!/bin/bash
# the default bench twelve times with the allocator's figures beside the throughput
OUT=gpurun_out/r3bn; mkdir -p $OUT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout -k 10 300 python bench.py --steps 16 --warmup 4 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['ms_per_step'], 'alloc', c['max_memory_allocated_gb'], 'reserved', c['max_memory_reserved_gb'], 'device_allocs', c['device_allocs'])" | tee -a $OUT/ab.txt
done
