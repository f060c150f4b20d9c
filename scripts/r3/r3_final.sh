#!/bin/bash
# final numbers of the round on one box: benches (each twice), kernel tables, GPU test tier
set -o pipefail
OUT=gpurun_out/r3final; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  timeout -k 10 300 python -u bench.py "$@" > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; return 1; }
  grep '^{' $OUT/b.log | tail -1 >> $OUT/bench.jsonl
  python - <<PY | tee -a $OUT/bench.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("$*:", d["value"], d["unit"], d["ms_per_step"], "ms mfu", d.get("mfu"), "vs_baseline", d.get("vs_baseline"), "loss", d["final_loss"])
PY
}
for rep in 1 2; do
run --steps 20 --warmup 5 || exit 1
run --batch 256 --steps 20 --warmup 5 || exit 1
run --model gpt2 --steps 20 --warmup 5 || exit 1
run --model gpt2 --batch 8 --steps 20 --warmup 5 || exit 1
run --model resnet152 --steps 8 --warmup 3 || exit 1
run --model resnet152 --batch 512 --steps 10 --warmup 3 || exit 1
run --model resnet152 --batch 256 --steps 10 --warmup 3 || exit 1
run --model mnist --steps 500 --warmup 50 || exit 1
done
run --model gpt2 --batch 32 --steps 12 --warmup 4 || exit 1
run --batch 1024 --steps 12 --warmup 4 || exit 1
export GPU_MAX_HW_QUEUES=8
bash scripts/gpu_profile.sh r50_r3final || exit 1
bash scripts/gpu_profile.sh gpt2_r3final --model gpt2 || exit 1
