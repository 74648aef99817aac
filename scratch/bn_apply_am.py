"""bn_apply with and without its magnitude by-product (round 4), on the tensors of the resnet-50 step"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspnet_amd import functional as fn
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for shape in [(32, 256, 256, 64), (32, 512, 512, 4), (32, 64, 64, 128), (32, 32, 32, 256), (32, 16, 16, 2048), (32, 64, 64, 172)]:
    x = torch.randn(*shape, device="cuda"); y = torch.empty_like(x)
    C = shape[-1]
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    am = torch.zeros(64, device="cuda")
    t0 = timeit(lambda: fn.bn_apply(x, sc, sh, relu=True, out=y))
    t1 = timeit(lambda: fn.bn_apply(x, sc, sh, relu=True, out=y, out_absmax=am))
    t2 = timeit(lambda: fn.absmax(y, out=am))
    gb = x.numel() * 8 / 1e9
    print(shape, "plain %.4f ms (%.2f TB/s) | +absmax %.4f ms | standalone absmax pass %.4f ms" % (t0, gb / t0, t1, t2))
    assert float(am.max()) == float(y.abs().max())
