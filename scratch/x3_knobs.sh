#!/bin/bash
# tile-selection knobs, one box: bash scratch/x3_knobs.sh
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline"
run() { echo -n "$1 $2: "; env $1 $B $2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run X=0 ""
run DSPN_NT_MINTILES=256 ""
run X=0 "--store bf16"
run DSPN_NT_MINTILES=256 "--store bf16"
run X=0 "--math fp32"
run DSPN_NT_MINTILES=256 "--math fp32"
done
run X=0 "--network inceptionv3 --size 512 --width 1024 --batch 8 --store bf16"
run DSPN_NT_MINTILES=256 "--network inceptionv3 --size 512 --width 1024 --batch 8 --store bf16"
run X=0 "--network vgg16_reduced --batch 16"
run DSPN_NT_MINTILES=256 "--network vgg16_reduced --batch 16"
