#!/bin/bash
# round-6 closing run on one box: the GPU suite (default math, then the two other fp32-result modes), the profile set (per-mode
# kernel statistics, last step by kernel, FETCH / WRITE traffic, MFMA busy), the layer table, the tile-spanning A/B, the
# inference line, the world-1 RCCL comparison and the default bench line.   usage: bash scratch/final_r06.sh [all|nosuite|noprofile|quick]
mkdir -p gpurun_out/final_r06 gpurun_out/profiles_r06
MODE=${1:-all}
if [ "$MODE" != "nosuite" ]; then
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final_r06/gpu_suite.log 2>&1; echo "suite rc $?"; grep -E "passed|failed" gpurun_out/final_r06/gpu_suite.log | tail -1
fi
if [ "$MODE" != "noprofile" ]; then
bash scratch/profile_round.sh r06 > gpurun_out/final_r06/profile.log 2>&1; echo "profile rc $?"
python scratch/layer_bench.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r06/r06_f16x2_layer_table.txt
python bench.py --mode infer --no-cpu-baseline > gpurun_out/profiles_r06/r06_infer_bench_line.json 2>/dev/null
bash scratch/r05/mgpu_r05.sh > gpurun_out/profiles_r06/r06_reserved_cus_world1.txt 2>&1
bash scratch/r05/ops_prof.sh > gpurun_out/final_r06/ops_prof.txt 2>&1; python scratch/r05/ops_split.py >> gpurun_out/final_r06/ops_prof.txt 2>&1
cp gpurun_out/final_r06/ops_prof.txt gpurun_out/profiles_r06/r06_multibox_ops_alone.txt
( echo "# same box, alternating: the tile-spanning loop off (DSPN_XT=0) / on (default)"; echo "# columns: images/s, ms per step, conv family TFLOP/s, conv family ms per step"
for R in 1 2 3; do for X in 0 1; do
DSPN_XT=$X python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('xt$X', l['value'], l['ms_per_step'], r['achieved'], r['conv_ms_per_step'])"
done; done ) > gpurun_out/profiles_r06/r06_tile_spanning_step_ab.txt 2>&1
( echo "# the closing run's box, alternating: DSPN_WGW=0 (conv_wgrad_kernel for the plane x plane weight gradients) / 1 (conv_wgw_kernel, default)"; bash scratch/r06/ab_env.sh DSPN_WGW 3 ) > gpurun_out/profiles_r06/r06_wgrad_wide_step_ab.txt 2>&1
( echo "# the closing run's box, alternating, with DSPN_WGRAD_SIDE=0 (everything on the step's stream): DSPN_FINALIZE_BESIDE=0 (the BatchNorm-backward finalize as launches of its own) / 1 (riding in front of the weight-gradient grid)"; DSPN_WGRAD_SIDE=0 bash scratch/r06/ab_env.sh DSPN_FINALIZE_BESIDE 3 ) > gpurun_out/profiles_r06/r06_finalize_rides_step_ab.txt 2>&1
( echo "# the closing run's box, alternating: DSPN_WGRAD_SIDE=0 (weight gradients on the step's stream, the BatchNorm finalize riding in them) / 1 (on a stream of their own beside the data-gradient chain, default)"; bash scratch/r06/ab_env.sh DSPN_WGRAD_SIDE 3 ) > gpurun_out/profiles_r06/r06_wgrad_beside_ab.txt 2>&1
fi
timeout 900 python bench.py > gpurun_out/final_r06/bench_default.log 2>&1; echo "bench rc $?"; tail -1 gpurun_out/final_r06/bench_default.log | cut -c1-600
tail -1 gpurun_out/final_r06/bench_default.log > gpurun_out/profiles_r06/r06_default_bench_line.json
if [ "$MODE" == "all" ]; then
for M in bf16x3 fp32; do
DSPN_CONV_MATH=$M timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final_r06/gpu_suite_$M.log 2>&1; echo "suite $M rc $?"; grep -E "passed|failed" gpurun_out/final_r06/gpu_suite_$M.log | tail -1
done
fi
