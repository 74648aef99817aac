#!/bin/bash
# round 5: wide-family parity tests, the per-layer table, and the same-box A/B of the training step (tiles automatic vs round-4 kernels)
OUT=gpurun_out/r05_step_ab; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wide_tiles_gpu.py -x -q -m gpu > $OUT/pytest_wide.log 2>&1; echo "pytest wide rc $?"; tail -3 $OUT/pytest_wide.log
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "piece_planes or f16x2 or two_piece" > $OUT/pytest_nn.log 2>&1; echo "pytest nn rc $?"; tail -2 $OUT/pytest_nn.log
python scratch/layer_bench.py 32 > $OUT/layer_table.txt 2>&1; tail -2 $OUT/layer_table.txt
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('conv_ms_per_step'))"; }
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --wide-tiles 1 2>/dev/null | tail -1 | line narrow
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line auto
done
for thr in 4 2 1; do
DSPN_X_PLANES_MIN_READS=$thr timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line "auto_minreads$thr"
done
