#!/bin/bash
# round 4: the detection branch's forward on a side stream beside the segmentation decoder (DSPN_DET_SIDE) -- A/B, then the graph tests
for i in 1 2 3; do
DSPN_DET_SIDE=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('off', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('on ', d['value'], d['ms_per_step'])"
done
timeout 1500 python -m pytest tests/test_graph_gpu.py -x -q -m gpu 2>&1 | grep -iE "passed|failed|error" | tail -3
