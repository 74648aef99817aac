#!/bin/bash
# round 4: MultiBoxTarget on a side stream beside the segmentation decoder's forward (DSPN_TARGET_SIDE) -- tests, then A/B
timeout 1500 python -m pytest tests/test_graph_gpu.py -x -q -m gpu 2>&1 | grep -iE "passed|failed|error" | tail -3
for i in 1 2 3; do
DSPN_TARGET_SIDE=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('off', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('on ', d['value'], d['ms_per_step'])"
done
