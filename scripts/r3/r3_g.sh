#!/bin/bash
# round-3 GPU pass G (re-entry baseline): whole GPU tier, the three benches, kernel tables of the ResNet-50 / GPT-2 steps
# (GPU_MAX_HW_QUEUES=8 exported in front of rocprofv3 so that profile and bench run on the same number of queues).
set -o pipefail
OUT=gpurun_out/r3g; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 > $OUT/bench_r50.log 2>&1 || { tail -20 $OUT/bench_r50.log; exit 1; }
tail -1 $OUT/bench_r50.log
timeout -k 10 200 python -u bench.py --model gpt2 --steps 20 --warmup 5 > $OUT/bench_gpt2.log 2>&1 || { tail -20 $OUT/bench_gpt2.log; exit 1; }
tail -1 $OUT/bench_gpt2.log
timeout -k 10 200 python -u bench.py --model resnet152 --steps 10 --warmup 3 > $OUT/bench_r152.log 2>&1 || { tail -20 $OUT/bench_r152.log; exit 1; }
tail -1 $OUT/bench_r152.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python3 bench.py --steps 4 --warmup 2 > $OUT/r50_prof.log 2>&1 || { tail -20 $OUT/r50_prof.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o gpt2 -- python3 bench.py --model gpt2 --steps 4 --warmup 2 > $OUT/gpt2_prof.log 2>&1 || { tail -20 $OUT/gpt2_prof.log; exit 1; }
find $OUT/prof -name "*kernel_stats*"
