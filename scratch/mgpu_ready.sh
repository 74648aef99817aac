# round 4, VERDICT item 5: what one GPU can say about the data-parallel run -- the RCCL path at world 1 with the persistent
# convolution grids leaving 0 / 8 / 16 CUs free, and the fields of the N > 1 line
python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "non_finite" 2>&1 | tail -2
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'exposed', d.get('allreduce_exposed_ms'), 'reserved', d.get('reserved_cus'), 'buckets', d.get('allreduce_bucket_latency_ms'), d['config'].get('f16x2_range_monitor'))"; }
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line plain
for k in 0 8 16; do
DSPN_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-other-configs --reserve-cus $k 2>/dev/null | tail -1 | line dist_reserve_$k
done
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line plain
