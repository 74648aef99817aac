"""affine sampler data gradient (+ theta rows) with and without the magnitude by-product, score3_conv's six components"""
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
th = torch.tensor([0.98, 0.03, -0.02, -0.04, 1.05, 0.01], device="cuda")
dy = torch.randn(32, 64, 64, 172, device="cuda")
for hw in (64, 32, 16, 8, 4, 2):
    x = torch.randn(32, hw, hw, 172, device="cuda")
    part = torch.zeros(fn.affine_sampler_theta_rows(x.shape, 64), 6, dtype=torch.float64, device="cuda")
    am = torch.zeros(64, device="cuda")
    dx = torch.empty_like(x)
    t0 = timeit(lambda: fn.affine_sampler_backward_data(dy, th, x.shape, 0, dx=dx))
    t1 = timeit(lambda: fn.affine_sampler_backward_data_theta(dy, th, x, 0, part, dx=dx))
    t2 = timeit(lambda: fn.affine_sampler_backward_data_theta(dy, th, x, 0, part, dx=dx, dx_absmax=am))
    t3 = timeit(lambda: fn.absmax(dx, out=am))
    print(hw, "data %.4f | data+theta %.4f | +absmax %.4f | standalone absmax %.4f ms" % (t0, t1, t2, t3))
