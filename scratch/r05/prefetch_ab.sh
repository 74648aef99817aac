#!/bin/bash
# round 5: deeper A-row prefetch of conv_ntv_kernel + pipelined epilogue reads of the wide family, against the build before them
# (dspnet_amd/libdspn_hip_var.so = HEAD~ library): parity tests of the new build, per-layer table of both, the step alternating
OUT=gpurun_out/r05_prefetch_ab; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wide_tiles_gpu.py -x -q -m gpu > $OUT/pytest_wide.log 2>&1; echo "pytest wide rc $?"; tail -3 $OUT/pytest_wide.log
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "piece_planes or f16x2 or two_piece" > $OUT/pytest_nn.log 2>&1; echo "pytest nn rc $?"; tail -2 $OUT/pytest_nn.log
python scratch/layer_bench.py 32 > $OUT/layer_table_new.txt 2>&1; tail -1 $OUT/layer_table_new.txt
DSPN_LIB=$PWD/dspnet_amd/libdspn_hip_var.so python scratch/layer_bench.py 32 > $OUT/layer_table_old.txt 2>&1; tail -1 $OUT/layer_table_old.txt
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('conv_ms_per_step'))"; }
for i in 1 2 3; do
DSPN_LIB=$PWD/dspnet_amd/libdspn_hip_var.so timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line old
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line new
done
