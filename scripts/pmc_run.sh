#!/bin/bash
# PMC counters of the kernels one script launches:  scripts/pmc_run.sh TAG "kernel filters" script.py [args]
# two passes (SQ instruction mix / SQ waits + LDS conflicts), each within the per-block counter limits;
# report -> gpurun_out/pmc/TAG/report.txt
set -o pipefail
TAG=$1; FILTERS=$2; SCRIPT=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc/$TAG
mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT ROUNDS=3
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o p1 -- python3 $GRAFT_REPO_ROOT/$SCRIPT "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o p2 -- python3 $GRAFT_REPO_ROOT/$SCRIPT "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
: > $OUT/report.txt
for f in $FILTERS; do python3 $GRAFT_REPO_ROOT/scripts/pmc_report.py $OUT $f >> $OUT/report.txt; done
sed -i "s#$GRAFT_REPO_ROOT/##" $OUT/report.txt
cat $OUT/report.txt
