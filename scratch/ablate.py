import sys
sys.path.insert(0, '/root/repo')
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
shapes = [(32, 32, 32, 256, 256, 3), (32, 64, 64, 128, 128, 3), (32, 16, 16, 512, 512, 3), (32, 32, 32, 1024, 256, 1)]
for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    res = []
    for bits in (0, 1, 3, 7, 15, 8):
        fn.L().dspn_debug_set(bits)
        t = timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out))
        res.append("dbg%-2d %.3fms %5.1fTF" % (bits, t, fl / t / 1e9))
    fn.L().dspn_debug_set(0)
    print((N, H, W, Cin, Cout, k), " | ".join(res))
