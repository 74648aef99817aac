#!/bin/bash
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline'].get('conv_ms_per_step'))"; }
for thr in 6 4 3 2 1; do
DSPN_X_PLANES_MIN_READS=$thr timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line "minreads$thr"
done
DSPN_X_PLANES_MIN_READS=6 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line "minreads6"
