#!/bin/bash
OUT=gpurun_out/r3cb; mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rows in 16384 8192; do timeout -k 10 120 python -u scripts/bench_ln.py $rows 768 2>&1 | grep -v amdgpu | tee -a $OUT/ln.txt; done
