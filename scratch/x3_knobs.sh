#!/bin/bash
# tile-selection knobs under the split math, one box: bash scratch/x3_knobs.sh
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-roofline"
run() { echo -n "$1: "; env $1 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run X=0
run DSPN_NT_MINTILES=256
run DSPN_NT_MINTILES=1024
run DSPN_NT_MINTILES=128
run DSPN_NT_8WAVE=0
run X=0
