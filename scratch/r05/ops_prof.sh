#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scratch/r05/ops_prof.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ops_prof -o k -- python3 scratch/r05/ops_prof.py > /dev/null 2>&1
head -14 gpurun_out/ops_prof/*kernel_stats.csv | cut -d, -f1-8
