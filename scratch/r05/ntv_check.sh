#!/bin/bash
OUT=gpurun_out/r05_ntv; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wide_tiles_gpu.py -x -q -m gpu > $OUT/wide.log 2>&1; echo "wide rc $?"; tail -4 $OUT/wide.log
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "conv or two_piece or f16x2" > $OUT/nn.log 2>&1; echo "nn rc $?"; tail -2 $OUT/nn.log
python scratch/layer_bench.py 32 > $OUT/layer_table.txt 2>&1; tail -1 $OUT/layer_table.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('auto  ', d['value'], d['ms_per_step'], d['roofline'].get('conv_ms_per_step'))"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --wide-tiles 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('narrow', d['value'], d['ms_per_step'], d['roofline'].get('conv_ms_per_step'))"
done
