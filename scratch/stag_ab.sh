# correctness of the f16x2 conv tests, then stagger on/off and the baseline library, bench line each (same box)
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "f16x2 or two_piece" 2>&1 | tail -3
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['achieved'])"; }
for i in 1 2; do
DSPN_LIB=dspnet_amd/libdspn_hip_base.so python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line base
DSPN_NT_NOSTAG=1 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line nostag
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line stag
done
