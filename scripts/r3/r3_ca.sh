#!/bin/bash
# the new rule (1024 blocks from 3M pixels) against 512 everywhere and 1024 everywhere, batch 1024, interleaved
OUT=gpurun_out/r3ca; mkdir -p $OUT
one() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 16 --warmup 4 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "$label $(grep -o '"value": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/ab.txt; }
for i in 1 2 3 4 5 6; do
  one rule X=1
  one all512 DPA_WGRAD_BLOCKS=512
  one all1024 DPA_WGRAD_BLOCKS=1024
done
python - <<'PY' | tee -a gpurun_out/r3ca/ab.txt
import collections
v = collections.defaultdict(list)
for line in open("gpurun_out/r3ca/ab.txt"):
    p = line.split()
    if len(p) == 3 and p[1] == '"value":': v[p[0]].append(float(p[2]))
for k, xs in v.items():
    print(k, "mean %.0f" % (sum(xs) / len(xs)), "min %.0f max %.0f" % (min(xs), max(xs)))
PY
