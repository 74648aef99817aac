#!/bin/bash
# LDS / MFMA / wait counters of one training step (two passes of <= 8 SQ counters): bash scratch/pmc_lds.sh <bench args, e.g. --store bf16 | --math bf16x3>
set -u
OUT=gpurun_out/pmc_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-other-configs $*"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -o p -- python3 bench.py $A > $OUT/a.log 2>&1
python3 scratch/pmc_any.py $OUT/a 20 > $OUT/a.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -o p -- python3 bench.py $A > $OUT/b.log 2>&1
python3 scratch/pmc_any.py $OUT/b 20 > $OUT/b.csv
find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/a.csv; cat $OUT/b.csv
