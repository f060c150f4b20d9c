#!/bin/bash
# PMC pass over the attention kernels at 16 x 12 x 1024 x 64: matrix-pipe busy cycles, vector / matrix co-execution, waits
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc/attn_r3
mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT -o p1 -- python3 $GRAFT_REPO_ROOT/scripts/bench_attn.py 16 > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o p2 -- python3 $GRAFT_REPO_ROOT/scripts/bench_attn.py 16 > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
: > $OUT/report.txt
for f in attn_fwd attn_bwd_q attn_bwd_kv; do python3 $GRAFT_REPO_ROOT/scripts/pmc_report.py $OUT $f >> $OUT/report.txt; done
sed -i "s#$GRAFT_REPO_ROOT/##" $OUT/report.txt
cat $OUT/report.txt
