#!/bin/bash
# do the vector ALU and the matrix pipe of the two-piece convolution kernels run beside each other?
# usage (GPU box): bash scratch/pmc_coexec.sh <tag> [bench args]   ->  gpurun_out/coexec_<tag>/{a,b}.csv
set -u
TAG=${1:-r04}; shift
OUT=gpurun_out/coexec_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
COMMON="--steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/a -o p -- python3 bench.py $COMMON "$@" > $OUT/a.log 2>&1
python3 scratch/pmc_any.py $OUT/a 24 > $OUT/a.csv; cat $OUT/a.csv | cut -c1-220
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM \
  --output-format csv -d $OUT/b -o p -- python3 bench.py $COMMON "$@" > $OUT/b.log 2>&1
python3 scratch/pmc_any.py $OUT/b 24 > $OUT/b.csv; cat $OUT/b.csv | cut -c1-220
find $OUT -name "*counter_collection.csv" -size +8M -delete; find $OUT -name "*.db" -delete
