#!/bin/bash
# round 4: part of the detection branch's BACKWARD on the side stream (DSPN_DET_SIDE_BWD=1) -- A/B, then the graph tests with it on
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwd only ', d['value'], d['ms_per_step'])"
DSPN_DET_SIDE_BWD=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwd + bwd', d['value'], d['ms_per_step'])"
done
export DSPN_DET_SIDE_BWD=1
timeout 1500 python -m pytest tests/test_graph_gpu.py -x -q -m gpu 2>&1 | grep -iE "passed|failed|error" | tail -3
