#!/bin/bash
# weight-gradient split target at the default batch 1024: 512 blocks (default) vs 768 vs 1024, paced bench, interleaved
OUT=gpurun_out/r3bz; mkdir -p $OUT
one() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 16 --warmup 4 > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  echo "$label $(grep -o '"value": [0-9.]*' $OUT/b.txt)" | tee -a $OUT/ab.txt; }
for i in 1 2 3 4 5 6; do
  one blocks=512 X=1
  one blocks=768 DPA_WGRAD_BLOCKS=768
  one blocks=1024 DPA_WGRAD_BLOCKS=1024
done
python - <<'PY' | tee -a gpurun_out/r3bz/ab.txt
import collections
v = collections.defaultdict(list)
for line in open("gpurun_out/r3bz/ab.txt"):
    p = line.split()
    if len(p) == 3 and p[0].startswith("blocks="): v[p[0]].append(float(p[2]))
for k, xs in v.items():
    print(k, "mean %.0f" % (sum(xs) / len(xs)), "min %.0f max %.0f" % (min(xs), max(xs)))
PY
