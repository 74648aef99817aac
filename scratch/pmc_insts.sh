#!/bin/bash
# instruction mix per kernel of one training step: bash scratch/pmc_insts.sh <bench args>
set -u
OUT=gpurun_out/pmc_insts
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-other-configs $*"
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH --output-format csv -d $OUT/a -o p -- python3 bench.py $A > $OUT/a.log 2>&1
python3 scratch/pmc_any.py $OUT/a 16 > $OUT/a.csv
find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/a.csv
