DSPN_NT_WIDE=1 timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -m gpu -k "f16x2 or two_piece" 2>&1 | tail -3
python scratch/layer_bench.py 32 > gpurun_out/lb_narrow.txt 2>&1
DSPN_NT_WIDE=128 python scratch/layer_bench.py 32 > gpurun_out/lb_wide.txt 2>&1
tail -1 gpurun_out/lb_narrow.txt; tail -1 gpurun_out/lb_wide.txt
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['achieved'])"; }
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line narrow
DSPN_NT_WIDE=128 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line wide128
DSPN_NT_WIDE=256 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line wide256
done
