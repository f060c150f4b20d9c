#!/bin/bash
# round-3 GPU pass F: halo conv correctness + A/B, then the rest of the tier, BN streaming A/B, rehearsal
set -o pipefail
OUT=gpurun_out/r3f; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_gpu_conv_halo.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest_halo.log 2>&1 || { tail -40 $OUT/pytest_halo.log; exit 1; }
tail -3 $OUT/pytest_halo.log
timeout -k 10 200 python -u scripts/bench_conv_halo.py --out $OUT/conv_halo_ab.json 2>&1 | tee $OUT/conv_halo_ab.txt || exit 1
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread --deselect tests/test_gpu_conv_halo.py > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
for cfg in "1 0 1" "4 0 1" "4 400 1" "4 0 0" "1 0 1" "4 0 1" "4 400 1" "4 0 0"; do
  set -- $cfg
  DPA_BN_UNROLL=$1 DPA_BN_NTS_MB=$2 DPA_CONV_HALO=$3 timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bn_unroll $1 nts_mb $2 conv_halo $3:', d['value'], d['ms_per_step'])" | tee -a $OUT/bench_ab.txt || exit 1
done
BATCH=512 DPA_BN_UNROLL=1 timeout -k 10 120 python -u scripts/bench_bn.py > $OUT/bn_u1.txt 2>&1 || exit 1
BATCH=512 DPA_BN_UNROLL=4 timeout -k 10 120 python -u scripts/bench_bn.py > $OUT/bn_u4.txt 2>&1 || exit 1
BATCH=512 DPA_BN_UNROLL=4 DPA_BN_NTS_MB=400 timeout -k 10 120 python -u scripts/bench_bn.py > $OUT/bn_u4_nts.txt 2>&1 || exit 1
bash scripts/rehearse_multirank.sh > $OUT/rehearse.log 2>&1 || { tail -30 $OUT/rehearse.log; exit 1; }
cat $OUT/rehearse.log
