#!/bin/bash
# SQ counters of the wide kernels on one layer (two passes): bash scratch/r05/pmc_ntw.sh <case> <dbg>
set -u
OUT=gpurun_out/r05_pmc_ntw_$1_$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o p -- scratch/r05/ntw_bench $1 $2 > $OUT/a.log 2>&1
python3 scratch/pmc_any.py $OUT/a 20 > $OUT/a.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/b -o p -- scratch/r05/ntw_bench $1 $2 > $OUT/b.log 2>&1
python3 scratch/pmc_any.py $OUT/b 20 > $OUT/b.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c -o p -- scratch/r05/ntw_bench $1 $2 > $OUT/c.log 2>&1
find $OUT -name "*counter_collection.csv" -size +8M -delete
cat $OUT/a.csv; cat $OUT/b.csv; cat $OUT/c/*kernel_stats.csv 2>/dev/null | cut -c1-200 | head -12
