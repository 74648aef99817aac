"""A/B of wgrad on one build: python ab_wgrad.py <lib.so | ->"""
import sys
sys.path.insert(0, '/root/repo')
from dspnet_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = sys.argv[1]
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
shapes = [(32, 16, 16, 512, 512, 3, 1), (32, 32, 32, 256, 256, 3, 1), (32, 64, 64, 128, 128, 3, 1), (32, 128, 128, 64, 64, 3, 1),
          (32, 64, 64, 128, 512, 1, 1), (32, 128, 128, 64, 256, 1, 1), (32, 128, 128, 256, 64, 1, 1), (32, 32, 32, 1024, 256, 1, 1),
          (32, 128, 128, 128, 128, 3, 2), (32, 64, 64, 256, 256, 3, 2), (32, 64, 64, 512, 1024, 1, 2), (32, 256, 256, 4, 64, 7, 2)]
tot = 0
for (N, H, W, Cin, Cout, k, st) in shapes:
    p = k // 2
    Ho = (H + 2 * p - k) // st + 1
    x = torch.randn(N, H, W, Cin, device=dev); dy = torch.randn(N, Ho, Ho, Cout, device=dev)
    sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev)
    fl = 2.0 * N * Ho * Ho * Cin * Cout * k * k
    t = timeit(lambda: fn.conv2d_wgrad(x, dy, (Cout, k, k, Cin), st, p, 1))
    t2 = timeit(lambda: fn.conv2d_wgrad(x, dy, (Cout, k, k, Cin), st, p, 1, in_affine=(sc, sh, True)))
    tot += t + t2
    print((N, H, W, Cin, Cout, k, st), "wgrad %.3f (%5.1fTF) | +affine %.3f (%5.1fTF)" % (t, fl / t / 1e9, t2, fl / t2 / 1e9))
print("total %.3f" % tot)
