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
shapes = [(32, 32, 32, 256, 256, 3), (32, 16, 16, 512, 512, 3), (32, 32, 32, 1024, 1024, 1), (32, 128, 128, 64, 256, 1), (32, 64, 64, 128, 512, 1)]
for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    by = 4.0 * N * H * W * (Cin + Cout)
    res = []
    for bits, nm in ((0, "full"), (16, "no-gstore"), (32, "no-epilogue"), (32 + 2 + 4, "no-epi,no-lds-store,no-barrier")):
        fn.L().dspn_debug_set(bits)
        t = timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out))
        res.append("%s %.3fms" % (nm, t))
    fn.L().dspn_debug_set(0)
    print((N, H, W, Cin, Cout, k), "MFMA-min %.3fms HBM-min %.3fms |" % (fl / 157e9, by / 5.8e9), " | ".join(res))
