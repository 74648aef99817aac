#!/bin/bash
# usage: bash scratch/prof_any.sh <tag> <bench.py args...>   -> gpurun_out/<tag>_last_step.txt
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline "$@" > $OUT/bench.log 2>&1
grep '^{"metric"' $OUT/bench.log | head -c 200; echo
python3 scratch/step_profile_csv.py $(ls $OUT/*kernel_trace.csv | head -1) 60 > gpurun_out/${TAG}_last_step.txt
rm -rf $OUT
