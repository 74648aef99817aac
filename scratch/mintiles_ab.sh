#!/bin/bash
# round 4: tile-count threshold of the 128x128 tile (stage 4 yields exactly 256 tiles = one workgroup per CU)
for i in 1 2; do
for T in 256 257 513; do
DSPN_NT_MINTILES=$T timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mintiles=$T', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
done
done
