#!/bin/bash
# SQ counters of one split-math layer: bash scratch/pmc_layer.sh <tag> <one_layer_x3.py args>; env passes through (DSPN_NT_NOHALO=1)
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_layer_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P() { d=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$d -o p -- python3 scratch/one_layer_x3.py $ARGS > $OUT/$d.log 2>&1; python3 scratch/pmc_any.py $OUT/$d 4 | grep -E "kernel|conv_" ; }
ARGS="$*"
P a SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS
P b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
P c SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_WAVES SQ_CYCLES
P d SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY
rocprofv3 --kernel-trace --stats -d $OUT/t -o p --output-format csv -- python3 scratch/one_layer_x3.py $ARGS > $OUT/t.log 2>&1
grep -h conv_ $OUT/t/*kernel_stats.csv $OUT/t/*/*kernel_stats.csv 2>/dev/null | cut -c1-60,200-400 | head -3
find $OUT -name "*counter_collection.csv" -size +8M -delete; find $OUT -name "*.db" -delete
