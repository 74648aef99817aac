#!/bin/bash
bash scratch/prof_any.sh r05_auto
bash scratch/prof_any.sh r05_narrow --wide-tiles 1
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --graph 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('eager', d['value'], d['ms_per_step'])"
