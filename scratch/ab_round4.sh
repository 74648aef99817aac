#!/bin/bash
# A/B of the working build against dspnet_amd/libdspn_hip_base.so on ONE box: correctness of the f16x2 conv tests first,
# then isolated layers and the bench line of both builds.   usage: bash scratch/ab_round4.sh <tag> [pytest -k expr]
TAG=${1:-ab}; K=${2:-"f16x2 or two_piece"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "$K" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
DSPN_LIB=dspnet_amd/libdspn_hip_base.so python scratch/layer_bench.py 32 > $OUT/lb_base.txt 2>&1
python scratch/layer_bench.py 32 > $OUT/lb_new.txt 2>&1
tail -1 $OUT/lb_base.txt; tail -1 $OUT/lb_new.txt
for i in 1 2; do
DSPN_LIB=dspnet_amd/libdspn_hip_base.so python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['value'], d['roofline']['achieved'])"
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['roofline']['achieved'])"
done
