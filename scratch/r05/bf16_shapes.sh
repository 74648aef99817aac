#!/bin/bash
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline'].get('conv_ms_per_step'))"; }
for m in 1 2 3 4 0 1; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --store bf16 --wide-tiles $m 2>/dev/null | tail -1 | line "bf16 tiles$m"
done
