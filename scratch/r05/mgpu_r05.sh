#!/bin/bash
# round 5, VERDICT r04 item 3: the RCCL path at world 1 WITH the side-stream backward active (plain vs dist within 1 %)
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'exposed', d.get('allreduce_exposed_ms'), 'reserved', d.get('reserved_cus'), 'side', d['config'].get('side_stream_schedule'), 'buckets', d.get('allreduce_bucket_latency_ms'))"; }
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line plain
for k in 0 16; do
DSPN_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-other-configs --reserve-cus $k 2>/dev/null | tail -1 | line dist_reserve_$k
done
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line plain
DSPN_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-other-configs --reserve-cus 0 2>/dev/null | tail -1 | line dist_reserve_0
