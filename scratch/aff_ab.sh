timeout 600 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "input_affine or batchnorm_statistics or full_size" 2>&1 | tail -2
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['achieved'], d['roofline']['nt_ms_per_step'])"; }
for i in 1 2; do
DSPN_LIB=dspnet_amd/libdspn_hip_prev.so python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line prev
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line new
done
