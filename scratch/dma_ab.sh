line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['achieved'])"; }
for i in 1 2; do
DSPN_LIB=dspnet_amd/libdspn_hip_nodma.so python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line regs
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | line dma
done
DSPN_LIB=dspnet_amd/libdspn_hip_nodma.so python scratch/layer_bench.py 32 > gpurun_out/lb_nodma.txt 2>&1
python scratch/layer_bench.py 32 > gpurun_out/lb_dma.txt 2>&1
tail -1 gpurun_out/lb_nodma.txt; tail -1 gpurun_out/lb_dma.txt
