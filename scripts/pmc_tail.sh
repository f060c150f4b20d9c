#!/bin/bash
# PMC counters of the fused bottleneck-tail backward kernel:  scripts/pmc_tail.sh TAG [N HW CM]
# three passes (SQ instruction mix / SQ waits + LDS conflicts / TCC traffic), each within the per-block counter limits
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc/$TAG
mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT ROUNDS=3
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o p1 -- python3 $GRAFT_REPO_ROOT/scripts/bench_tail.py "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o p2 -- python3 $GRAFT_REPO_ROOT/scripts/bench_tail.py "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o p3 -- python3 $GRAFT_REPO_ROOT/scripts/bench_tail.py "$@" > $OUT/p3.log 2>&1 || { tail -5 $OUT/p3.log; exit 1; }
python3 $GRAFT_REPO_ROOT/scripts/pmc_report.py $OUT tail_bwd_kernel > $OUT/report_tail.txt
python3 $GRAFT_REPO_ROOT/scripts/pmc_report.py $OUT bn_bwd_apply >> $OUT/report_tail.txt
cat $OUT/report_tail.txt
