#!/bin/bash
# round 4: BatchNorm backward with the pooled gradient formed on the fly (DSPN_FUSE_POOL_BWD) -- tests, then the bench line with / without
timeout 600 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "pooled_gradient or maxpool" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 1500 python -m pytest tests/test_graph_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
DSPN_FUSE_POOL_BWD=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('off', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('on ', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
done
