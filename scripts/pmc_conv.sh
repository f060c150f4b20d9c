#!/bin/bash
# PMC counters of one conv kernel config:  scripts/pmc_conv.sh TAG H C K R stride pad {fprop|dgrad|wgrad}
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o p1 -- python3 $GRAFT_REPO_ROOT/scripts/one_conv.py "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 90 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p2 -- python3 $GRAFT_REPO_ROOT/scripts/one_conv.py "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
ls $OUT
