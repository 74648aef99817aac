#!/bin/bash
# round 4: threshold of the input piece planes (reads per element at which a deferred BatchNorm also writes planes)
for i in 1 2 3; do
for T in 9 8 6 5 4; do
DSPN_X_PLANES_MIN_READS=$T timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('T=$T', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
done
done
