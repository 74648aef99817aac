#!/bin/bash
# round 4: is the pre-reduction of long tile tables (tile_group_kernel, 49 launches) worth its launches?
for i in 1 2; do
for T in 1024 100000000; do
DSPN_TILE_GROUP_MIN=$T timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('T=$T', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
done
done
