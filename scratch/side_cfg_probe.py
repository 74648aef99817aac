"""why is a side measurement of bench.py slower than the same configuration run alone?  fp32 alone, then f16x2 followed by fp32"""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
order = sys.argv[1:] or ["fp32"]
for m in order:
    r = bench.side_train("resnet-50", 512, 512, 32, m, 6, 2, dev)
    print(m, r["ms_per_step"], "conv ms", r["roofline"].get("conv_ms_per_step"), "launches", r["roofline"].get("launches_per_step"), flush=True)
    gc.collect(); torch.cuda.empty_cache()
