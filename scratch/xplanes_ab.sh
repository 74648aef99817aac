#!/bin/bash
# round 4: input piece planes (DSPN_X_PLANES) -- the functional test, the graph parity tests, then the bench line with / without
OUT=gpurun_out/xplanes; mkdir -p $OUT
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "piece_planes" > $OUT/pytest_nn.log 2>&1; echo "nn rc $?"; tail -5 $OUT/pytest_nn.log
timeout 1500 python -m pytest tests/test_graph_gpu.py -x -q -m gpu > $OUT/pytest_graph.log 2>&1; echo "graph rc $?"; tail -5 $OUT/pytest_graph.log
for i in 1 2; do
DSPN_X_PLANES=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('off', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('on ', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
done
