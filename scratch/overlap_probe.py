"""can an HBM-bound elementwise kernel (bn_apply: 30 VGPRs) run INSIDE a convolution launch of another stream?  times a weight
gradient loop and a bn_apply loop alone and together on two streams"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
N, H, W, C = 32, 32, 32, 256
x = torch.randn(N, H, W, C, device=dev); dy = torch.randn(N, H, W, C, device=dev)
aff = (torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev), True)
wshape = (C, 3, 3, C)
dw = torch.empty(wshape, device=dev)
big = torch.randn(32, 64, 64, 256, device=dev); outb = torch.empty_like(big)
sc, sh = torch.rand(256, device=dev), torch.rand(256, device=dev)
def conv(): fn.conv2d_wgrad(x, dy, wshape, 1, 1, 1, out=dw, in_affine=aff)
def elem(): fn.bn_apply(big, sc, sh, relu=True, out=outb)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(fa, fb, reps=20):
    for f, s in ((fa, sa), (fb, sb)):
        if f:
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_event(e0); sb.wait_event(e0)
    for _ in range(reps):
        if fa:
            with torch.cuda.stream(sa): fa()
        if fb:
            with torch.cuda.stream(sb): fb()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa); eb.record(sb)
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mode in ("bf16x3", "fp32"):
    fn.set_conv_math(mode)
    for rep in range(3):
      ta, tb, tab = run(conv, None, 50), run(None, elem, 50), run(conv, elem, 50)
      print(mode, "wgrad alone %.3f ms | bn_apply alone %.3f ms | both streams %.3f ms (sum %.3f, max %.3f)" % (ta, tb, tab, ta + tb, max(ta, tb)))
