#!/bin/bash
# round 6: the tile-spanning loop (DSPN_XT=1, default) against the round-5 loop (DSPN_XT=0) on one box: layer table and step
mkdir -p gpurun_out/r06
for X in 0 1; do
DSPN_XT=$X python scratch/layer_bench.py 32 > gpurun_out/r06/layers_xt$X.txt 2>&1
done
for R in 1 2 3; do for X in 0 1; do
DSPN_XT=$X python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l['roofline']
print('xt$X', l['value'], l['ms_per_step'], r['achieved'], r['conv_ms_per_step'])"
done; done > gpurun_out/r06/step_xt_ab.txt 2>&1
cat gpurun_out/r06/step_xt_ab.txt
