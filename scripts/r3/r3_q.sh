#!/bin/bash
# wide weight gradients in the GPT-2 step: A/B (off / per layer / grouped), then the GPU test tier
set -o pipefail
OUT=gpurun_out/r3q; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
one() {  # label, env..., -- bench args
  local label=$1; shift
  env "$@" timeout -k 10 300 python -u bench.py --model gpt2 --steps 12 --warmup 4 --batch $B > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; return 1; }
  python - <<PY | tee -a $OUT/ab.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("$label batch $B:", d["value"], d["ms_per_step"], "loss", d["final_loss"])
PY
}
for rep in 1 2; do
for B in 16 32; do
  one "wide off      " DPA_WGRAD_WIDE=0 || exit 1
  one "wide per layer" DPA_WGRAD_GROUP=0 || exit 1
  one "wide grouped  " DPA_WGRAD_GROUP=1 || exit 1
done; done
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $OUT/pytest.txt 2>&1; rc=$?
tail -8 $OUT/pytest.txt
exit $rc
