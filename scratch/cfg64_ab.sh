#!/bin/bash
# round 4: the Cout = 64 layers (conv0, stage 1) on 128 x 64 tiles (DSPN_NT_CFG64=1) instead of 64 x 64
python scratch/layer_bench.py 32 2>/dev/null | grep -E "^layer|conv0|stage1" > gpurun_out/cfg64_base.txt
DSPN_NT_CFG64=1 python scratch/layer_bench.py 32 2>/dev/null | grep -E "conv0|stage1" > gpurun_out/cfg64_c1.txt
cat gpurun_out/cfg64_base.txt; echo "--- cfg 1 (128x64)"; cat gpurun_out/cfg64_c1.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
DSPN_NT_CFG64=1 timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg1', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
done
