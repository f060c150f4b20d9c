#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3w; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  timeout -k 10 300 python -u bench.py "$@" > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; return 1; }
  python - <<PY | tee -a $OUT/bench.txt
import json
d = json.loads([l for l in open("$OUT/b.log") if l.startswith("{")][-1])
print("$*:", d["value"], d["unit"], d["ms_per_step"], "ms mfu", d.get("mfu"), "loss", d["final_loss"])
PY
}
run --model gpt2 --batch 16 --steps 12 --warmup 4 || exit 1
run --model gpt2 --batch 32 --steps 12 --warmup 4 || exit 1
run --model gpt2 --batch 8 --steps 12 --warmup 4 || exit 1
run --model gpt2 --batch 16 --steps 12 --warmup 4 || exit 1
run --steps 12 --warmup 4 || exit 1
run --batch 256 --steps 12 --warmup 4 || exit 1
run --model resnet152 --steps 8 --warmup 3 || exit 1
run --model mnist --steps 300 --warmup 50 || exit 1
timeout -k 10 600 python -u -m pytest tests/test_gpu_gemm_wide.py tests/test_gpu_wgrad_wide.py tests/test_gpu_e2e.py tests/test_gpu_real_shapes.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -5 | tee $OUT/pytest.txt
