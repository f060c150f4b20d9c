#!/bin/bash
# run-ahead depth against speed and against the slow mode: 0 (unbounded), 2, 4, 8, interleaved, twelve rounds, default bench
OUT=gpurun_out/r3bp; mkdir -p $OUT
one() { timeout -k 10 300 python bench.py --steps 16 --warmup 4 "$@" > $OUT/b.txt 2>&1 || { tail -5 $OUT/b.txt; exit 1; }
  grep '^{' $OUT/b.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('run_ahead', c['host_run_ahead_steps'], d['value'], d['ms_per_step'], 'reserved', c['max_memory_reserved_gb'])" | tee -a $OUT/ab.txt; }
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  for d in 0 2 4 8; do one --run-ahead $d; done
done
python - <<'PY' | tee -a gpurun_out/r3bp/ab.txt
import collections
v = collections.defaultdict(list)
for line in open("gpurun_out/r3bp/ab.txt"):
    p = line.split()
    if p and p[0] == "run_ahead": v[int(p[1])].append(float(p[2]))
for d, xs in sorted(v.items()):
    ok = [x for x in xs if x > 12000]
    print(f"depth {d}: {len(xs)} runs, {len(xs) - len(ok)} slow, fast-run mean {sum(ok) / max(1, len(ok)):.0f} img/s, min {min(xs):.0f}")
PY
