"""A/B of one libdspn_hip.so build on fwd / residual fwd / dgrad (stride 1 and 2, with and without accumulate)"""
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
# (N, H, W, Cin, Cout, k, stride)
shapes = [(32, 16, 16, 512, 512, 3, 1), (32, 32, 32, 256, 256, 3, 1), (32, 64, 64, 128, 128, 3, 1), (32, 128, 128, 64, 64, 3, 1),
          (32, 64, 64, 128, 512, 1, 1), (32, 128, 128, 64, 256, 1, 1), (32, 128, 128, 256, 64, 1, 1), (32, 64, 64, 512, 128, 1, 1),
          (32, 128, 128, 256, 512, 1, 2), (32, 64, 64, 512, 1024, 1, 2), (32, 128, 128, 128, 128, 3, 2), (32, 64, 64, 256, 256, 3, 2)]
tot = [0, 0, 0, 0]
for (N, H, W, Cin, Cout, k, st) in shapes:
    p = k // 2
    Ho = (H + 2 * p - k) // st + 1
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, Ho, Ho, Cout, device=dev); res = torch.randn(N, Ho, Ho, Cout, device=dev)
    dy = torch.randn(N, Ho, Ho, Cout, device=dev); dx = torch.zeros(N, H, W, Cin, device=dev)
    wt = fn.weight_transpose(w)
    fl = 2.0 * N * Ho * Ho * Cin * Cout * k * k
    t = [timeit(lambda: fn.conv2d_forward(x, w, None, st, p, 1, out=out)),
         timeit(lambda: fn.conv2d_forward(x, w, None, st, p, 1, out=out, residual=res)),
         timeit(lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), st, p, 1, out=dx)),
         timeit(lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), st, p, 1, out=dx, accumulate=True))]
    for i in range(4): tot[i] += t[i]
    print((N, H, W, Cin, Cout, k, st), "fwd %.3f (%5.1fTF) | +res %.3f | dgrad %.3f (%5.1fTF) | +acc %.3f" % (t[0], fl / t[0] / 1e9, t[1], t[2], fl / t[2] / 1e9, t[3]))
print("totals fwd %.3f +res %.3f dgrad %.3f +acc %.3f" % tuple(tot))
