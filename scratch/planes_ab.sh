python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "piece_planes_from_batchnorm or backward_sums" 2>&1 | tail -2
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['achieved'])"; }
for i in 1 2; do
DSPN_DY_PLANES=0 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line floats
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line planes
done
