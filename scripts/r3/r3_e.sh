#!/bin/bash
# round-3 GPU pass: test tier, BatchNorm streaming A/B (unroll / grid / non-temporal stores), multi-rank rehearsal
set -o pipefail
OUT=gpurun_out/r3e; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
for cfg in "1 0" "4 0" "4 400"; do
  set -- $cfg
  BATCH=512 DPA_BN_UNROLL=$1 DPA_BN_NTS_MB=$2 timeout -k 10 120 python -u scripts/bench_bn.py > $OUT/bn_u$1_nts$2.txt 2>&1 || { tail $OUT/bn_u$1_nts$2.txt; exit 1; }
done
for cfg in "1 0" "4 0" "4 400" "1 0" "4 0" "4 400"; do
  set -- $cfg
  DPA_BN_UNROLL=$1 DPA_BN_NTS_MB=$2 timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unroll $1 nts_mb $2:', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_bn_ab.txt || exit 1
done
bash scripts/rehearse_multirank.sh > $OUT/rehearse.log 2>&1 || { tail -30 $OUT/rehearse.log; exit 1; }
cat $OUT/rehearse.log
