#!/bin/bash
# fused cross-entropy: 512-thread workgroups (two per CU) vs 1024-thread (one per CU) on GPT-2's 50,304-column rows
OUT=gpurun_out/r3bf; mkdir -p $OUT
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_real_shapes.py tests/test_gpu_e2e.py -x -q -k "ce or cross or loss or gpt2 or lm_head or lmhead" --timeout 120 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for th in 512 1024; do
  DPA_CE_THREADS=$th timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p$th -o p -- python3 $GRAFT_REPO_ROOT/bench.py --model gpt2 --steps 6 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/p$th.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$OUT/p$th.log; exit 1; }
  echo "threads=$th $(grep ce_fused $GRAFT_REPO_ROOT/$OUT/p$th/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,120-)" | tee -a $GRAFT_REPO_ROOT/$OUT/ab.txt
done
cd $GRAFT_REPO_ROOT
for th in 512 1024 512 1024; do
  DPA_CE_THREADS=$th timeout -k 10 300 python bench.py --model gpt2 --steps 30 --warmup 10 > $OUT/g.txt 2>&1 || { tail -5 $OUT/g.txt; exit 1; }
  echo "threads=$th $(grep -o '"value": [0-9.]*' $OUT/g.txt)" | tee -a $OUT/ab.txt
done
