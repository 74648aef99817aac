#!/bin/bash
# same-box A/B of arbitrary values of one environment variable: bash scratch/r06/ab_envv.sh VAR "v1 v2 ..." [rounds]
V=$1; S=$2; R=${3:-3}
echo "# columns: setting, images/s, ms per step, conv family TFLOP/s, conv family ms per step"
for r in $(seq $R); do for X in $S; do
env $V=$X python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('$V=$X', l['value'], l['ms_per_step'], r['achieved'], r['conv_ms_per_step'])"
done; done
