#!/bin/bash
# same-box A/B of DSPN_WGRAD_SIDE on the other training configurations: vgg16_reduced bs 16 (configs[1]), inceptionv3 1024x512 bs 8 bf16 (configs[3]),
# the headline shape with bf16 tensors.   usage: bash scratch/r06/ab_other.sh [rounds]
for r in $(seq ${1:-2}); do for X in 0 1; do
for CFG in "--network vgg16_reduced --batch 16" "--network inceptionv3 --size 512 --width 1024 --batch 8 --math bf16 --store bf16" "--store bf16 --math bf16"; do
env DSPN_WGRAD_SIDE=$X python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs --no-roofline --sustained-steps 0 $CFG 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('DSPN_WGRAD_SIDE=$X', '$CFG', l['value'], l['ms_per_step'])"
done; done; done
