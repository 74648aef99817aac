#!/bin/bash
# same-box A/B of the working build against an experiment build (make VARIANT_FLAGS=... -> libdspn_hip_var.so):
# planes tests on the variant, the data-gradient layer table of both, the bench line of both twice
OUT=gpurun_out/var_ab; mkdir -p $OUT
DSPN_LIB=dspnet_amd/libdspn_hip_var.so timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "piece_planes or f16x2 or two_piece" > $OUT/pytest.log 2>&1; echo "pytest(var) rc $?"; tail -2 $OUT/pytest.log
echo base; python scratch/planes_layer.py 2>/dev/null
echo var; DSPN_LIB=dspnet_amd/libdspn_hip_var.so python scratch/planes_layer.py 2>/dev/null
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
DSPN_LIB=dspnet_amd/libdspn_hip_var.so timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('var ', d['value'], d['roofline']['achieved'], d['ms_per_step'])"
done
