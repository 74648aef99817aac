python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "affine_sampler or batchnorm" 2>&1 | tail -3
python -m pytest tests/test_graph_gpu.py -x -q -m gpu -k "f16x2 and resnet" 2>&1 | tail -4
python scratch/absmax_trace.py 32 2>&1 | tee gpurun_out/absmax_trace.txt | head -45
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_tail -o kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline > gpurun_out/kt_tail.log 2>&1
python3 scratch/step_profile_csv.py $(ls gpurun_out/kt_tail/*kernel_trace.csv | head -1) 90 > gpurun_out/last_step_tail.txt; head -4 gpurun_out/last_step_tail.txt; grep -E "absmax|sampler|target_match|bn_apply" gpurun_out/last_step_tail.txt
find gpurun_out/kt_tail -name "*.csv" -size +8M -delete
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | cut -c1-200
