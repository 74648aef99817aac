#!/bin/bash
# round-5 closing run on one box: the whole GPU suite (default math, then the two other fp32-result modes), the profile set
# (per-mode kernel statistics, last step by kernel, FETCH / WRITE traffic, MFMA busy), the layer table, the inference line,
# the world-1 RCCL comparison and the default bench line
mkdir -p gpurun_out/final_r05 gpurun_out/profiles_r05
if [ "${1:-all}" != "nosuite" ]; then
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final_r05/gpu_suite.log 2>&1; echo "suite rc $?"; tail -2 gpurun_out/final_r05/gpu_suite.log
fi
if [ "${1:-all}" != "noprofile" ]; then
bash scratch/profile_round.sh r05 > gpurun_out/final_r05/profile.log 2>&1; echo "profile rc $?"
python scratch/layer_bench.py 32 > gpurun_out/profiles_r05/r05_f16x2_layer_table.txt 2>&1
python bench.py --mode infer --no-cpu-baseline > gpurun_out/profiles_r05/r05_infer_bench_line.json 2>/dev/null
bash scratch/r05/mgpu_r05.sh > gpurun_out/profiles_r05/r05_reserved_cus_world1.txt 2>&1
bash scratch/r05/ops_prof.sh > gpurun_out/final_r05/ops_prof.txt 2>&1; python scratch/r05/ops_split.py >> gpurun_out/final_r05/ops_prof.txt 2>&1
cp gpurun_out/final_r05/ops_prof.txt gpurun_out/profiles_r05/r05_multibox_ops_alone.txt
fi
timeout 900 python bench.py > gpurun_out/final_r05/bench_default.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/final_r05/bench_default.log | cut -c1-600
tail -1 gpurun_out/final_r05/bench_default.log > gpurun_out/profiles_r05/r05_default_bench_line.json
if [ "${1:-all}" == "all" ]; then
for M in bf16x3 fp32; do
DSPN_CONV_MATH=$M timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final_r05/gpu_suite_$M.log 2>&1; echo "suite $M rc $?"; tail -2 gpurun_out/final_r05/gpu_suite_$M.log
done
fi
