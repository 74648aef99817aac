DSPN_LIB=dspnet_amd/libdspn_hip_wg2.so timeout 600 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "f16x2 and (forward_dgrad_wgrad or input_affine or full_size or planes)" 2>&1 | tail -2
DSPN_LIB=dspnet_amd/libdspn_hip_wg2.so DSPN_DEBUG_PRINT=1 python scratch/wg_one.py 32 32 32 256 256 3 2>&1 | tail -3
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['achieved'], d['roofline']['wgrad_ms_per_step'])"; }
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line one_stage
DSPN_LIB=dspnet_amd/libdspn_hip_wg2.so python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line two_stages
done
