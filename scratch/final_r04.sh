#!/bin/bash
# round-4 closing run on one box: the whole GPU suite, the profile set, the default bench line
mkdir -p gpurun_out/final_r04
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final_r04/gpu_suite.log 2>&1; echo "suite rc $?"; tail -2 gpurun_out/final_r04/gpu_suite.log
bash scratch/profile_r04.sh > gpurun_out/final_r04/profile.log 2>&1; echo "profile rc $?"
timeout 900 python bench.py > gpurun_out/final_r04/bench_default.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/final_r04/bench_default.log | cut -c1-400
