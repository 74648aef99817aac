#!/bin/bash
# round-5 closing run on one box: the whole GPU suite, the profile set (per-mode kernel statistics, last step by kernel,
# FETCH / WRITE traffic, MFMA busy), the layer table, the inference line and the default bench line
mkdir -p gpurun_out/final_r05
if [ "${1:-all}" != "noprofile" ]; then
bash scratch/profile_round.sh r05 > gpurun_out/final_r05/profile.log 2>&1; echo "profile rc $?"
python scratch/layer_bench.py 32 > gpurun_out/profiles_r05/r05_f16x2_layer_table.txt 2>&1
python bench.py --mode infer --no-cpu-baseline > gpurun_out/profiles_r05/r05_infer_bench_line.json 2>/dev/null
fi
if [ "${1:-all}" != "nosuite" ]; then
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final_r05/gpu_suite.log 2>&1; echo "suite rc $?"; tail -2 gpurun_out/final_r05/gpu_suite.log
fi
timeout 900 python bench.py > gpurun_out/final_r05/bench_default.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/final_r05/bench_default.log | cut -c1-600
