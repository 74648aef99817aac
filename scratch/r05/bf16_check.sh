#!/bin/bash
OUT=gpurun_out/r05_bf16; mkdir -p $OUT
timeout 600 python -m pytest tests/test_wide_tiles_gpu.py -x -q -m gpu > $OUT/wide.log 2>&1; echo "wide rc $?"; tail -4 $OUT/wide.log
timeout 900 python -m pytest tests/test_bf16_storage_gpu.py -x -q -m gpu > $OUT/bf16.log 2>&1; echo "bf16 rc $?"; tail -3 $OUT/bf16.log
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('conv_ms_per_step'))"; }
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --store bf16 2>/dev/null | tail -1 | line "bf16 auto  "
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --store bf16 --wide-tiles 1 2>/dev/null | tail -1 | line "bf16 narrow"
done
