"""how fast do two fp32 evaluations of the same 5-step training run drift apart?  control for the split-math comparison:
fused vs stand-alone BatchNorm kernels (both fp32 MFMA), and bf16x3 vs fp32 MFMA"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from dspnet_amd import engine as E, functional as fn
from dspnet_amd.train.metric import MultiBoxMetric
from test_graph_gpu import make
def run(math, fuse=True, size=256, batch=2):
    fn.set_conv_math(math); E.FUSE_BATCHNORM = fuse
    try:
        net, solver, *_ = make(batch, size, size)
    finally:
        E.FUSE_BATCHNORM = True
    m, hist = MultiBoxMetric(), []
    for _ in range(5):
        solver.step(); m.reset(); m.update(net); hist.append(m.get()[1])
    torch.cuda.synchronize()
    return np.asarray(hist), net.g.arena.double().clone()
for size, batch in ((256, 2), (512, 4)):
    hf, af = run("fp32", size=size, batch=batch); hu, au = run("fp32", fuse=False, size=size, batch=batch); hs, as_ = run("bf16x3", size=size, batch=batch)
    print(size, batch, "fp32 fused vs fp32 unfused:", np.abs(hu / hf - 1).max(axis=0), "params %.2e" % float((au - af).norm() / af.norm()))
    print(size, batch, "bf16x3 vs fp32 (both fused):", np.abs(hs / hf - 1).max(axis=0), "params %.2e" % float((as_ - af).norm() / af.norm()))
    print("   per step bf16x3/fp32:", np.abs(hs / hf - 1).tolist())
