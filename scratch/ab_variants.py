"""time a fixed conv_nt layer set with one build: python ab_variants.py <lib.so>"""
import sys
sys.path.insert(0, '/root/repo')
from dspnet_amd import _lib
_lib.LIB_PATH = sys.argv[1]
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
shapes = [(32, 16, 16, 512, 512, 3), (32, 32, 32, 256, 256, 3), (32, 64, 64, 128, 128, 3), (32, 128, 128, 64, 64, 3), (32, 32, 32, 1024, 256, 1), (32, 32, 32, 256, 1024, 1), (32, 64, 64, 128, 512, 1), (32, 128, 128, 64, 256, 1)]
out = []
tot = 0
for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    o = torch.empty(N, H, W, Cout, device=dev)
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    t = timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=o))
    tot += t; out.append("%.3f(%3.0f)" % (t, fl / t / 1e9))
print(sys.argv[1].split('/')[-1], "total %.3f ms | " % tot, " ".join(out))
