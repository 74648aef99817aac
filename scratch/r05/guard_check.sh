#!/bin/bash
OUT=gpurun_out/r05_guard; mkdir -p $OUT
timeout 900 python -m pytest tests/test_wide_tiles_gpu.py -x -q -m gpu > $OUT/wide.log 2>&1; echo "wide rc $?"; tail -3 $OUT/wide.log
timeout 1200 python -m pytest tests/test_graph_gpu.py -x -q -m gpu -s -k "channel_spread or reducer_path or training_reduces or two_backward or hip_graph" > $OUT/graph.log 2>&1; echo "graph rc $?"; grep -E "^(fp32|bf16x3|f16x2)|passed|failed|Error" $OUT/graph.log | cut -c1-400 | tail -12
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "piece_planes or planes or batchnorm or bn_" > $OUT/nn.log 2>&1; echo "nn rc $?"; tail -2 $OUT/nn.log
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('guard on ', d['value'], d['ms_per_step'], d['config'].get('f16x2_fallback_calls'), d['config'].get('f16x2_range_guard'))"
DSPN_RANGE_GUARD=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('guard off', d['value'], d['ms_per_step'])"
done
