#!/bin/bash
# First-contact GPU run: kernel numerics, smoke, short bench, stock baseline.
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
rc=$?; cat gpurun_out/pytest_gpu.log | tail -25
[ $rc -eq 0 ] || { echo "pytest failed rc=$rc"; exit $rc; }
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log || exit 1
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 2>&1 | tail -5 | tee gpurun_out/bench_r50.log || exit 1
timeout -k 10 300 python -u scripts/baseline_torch.py --steps 10 --warmup 3 --out gpurun_out/baseline_torch.json 2>&1 | tail -3 | tee gpurun_out/baseline.log || exit 1
