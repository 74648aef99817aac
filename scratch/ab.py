import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
bits = [int(b) for b in sys.argv[1:]] or [0, 64]
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
shapes = [(32, 16, 16, 512, 512, 3), (32, 32, 32, 256, 256, 3), (32, 32, 32, 1024, 256, 1), (32, 64, 64, 128, 128, 3), (32, 64, 64, 128, 512, 1), (32, 128, 128, 64, 64, 3), (32, 128, 128, 64, 256, 1)]
for rnd in range(2):
  for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    res = []
    for b in bits:
        fn.L().dspn_debug_set(b)
        t = timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out))
        res.append("dbg%-3d %.3fms %5.1fTF" % (b, t, fl / t / 1e9))
    fn.L().dspn_debug_set(0)
    print((N, H, W, Cin, Cout, k), " | ".join(res))
