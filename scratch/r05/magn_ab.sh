#!/bin/bash
# round 5: convolution epilogues / the ReLU-backward pass leave the operand magnitudes of the next two-piece calls
# (DSPN_CONV_MAGNITUDES=0: the stand-alone passes), same box, alternating; vgg16_reduced bs 16 and the headline graph
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
DSPN_CONV_MAGNITUDES=0 python bench.py --network vgg16_reduced --batch 16 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line vgg_passes
python bench.py --network vgg16_reduced --batch 16 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line vgg_fused
done
for i in 1 2 3; do
DSPN_CONV_MAGNITUDES=0 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line r50_passes
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line r50_fused
done
