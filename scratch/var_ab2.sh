#!/bin/bash
# same-box A/B of libdspn_hip.so against libdspn_hip_var.so: fused-epilogue costs, then the bench line, twice each
echo base; python scratch/fuse_cost.py 2>/dev/null | tail -10
echo var; DSPN_LIB=dspnet_amd/libdspn_hip_var.so python scratch/fuse_cost.py 2>/dev/null | tail -10
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
DSPN_LIB=dspnet_amd/libdspn_hip_var.so timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('var ', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
done
