"""per-layer time of this library's convolution kernels next to torch's ROCm convolution (MIOpen, channels_last and
NCHW, after its own algorithm search) on the resnet-50 shapes of the benchmark (batch 32, 512x512 input)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import torch.nn.functional as F
from dspnet_amd import functional as fn
torch.backends.cudnn.benchmark = True          # lets MIOpen search for its fastest algorithm per shape
dev = torch.device("cuda", 0)
def timeit(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
shapes = [(32, 128, 128, 64, 64, 3, 1, 1), (32, 128, 128, 256, 64, 1, 1, 0), (32, 128, 128, 64, 256, 1, 1, 0),
          (32, 64, 64, 128, 128, 3, 1, 1), (32, 64, 64, 512, 128, 1, 1, 0), (32, 32, 32, 256, 256, 3, 1, 1),
          (32, 32, 32, 1024, 256, 1, 1, 0), (32, 16, 16, 512, 512, 3, 1, 1), (32, 128, 128, 256, 128, 3, 2, 1)]
tot = {"ours": 0.0, "cl": 0.0, "nchw": 0.0}
print("%-38s | %-26s | %-26s | %-26s" % ("N,H,W,Cin,Cout,k,s,p", "ours fwd/dgrad/wgrad ms", "torch channels_last", "torch NCHW"))
for (N, H, W, Cin, Cout, k, st, pd) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    y = fn.conv2d_forward(x, w, None, st, pd, 1)
    dy = torch.randn_like(y)
    wt = fn.weight_transpose(w)
    o = [timeit(lambda: fn.conv2d_forward(x, w, None, st, pd, 1, out=y)),
         timeit(lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), st, pd, 1)),
         timeit(lambda: fn.conv2d_wgrad(x, dy, tuple(w.shape), st, pd, 1))]
    res = {}
    for name, fmt in (("cl", torch.channels_last), ("nchw", torch.contiguous_format)):
        xr = x.permute(0, 3, 1, 2).contiguous(memory_format=fmt)
        wr = w.permute(0, 3, 1, 2).contiguous(memory_format=fmt)
        dyr = dy.permute(0, 3, 1, 2).contiguous(memory_format=fmt)
        f0 = timeit(lambda: F.conv2d(xr, wr, None, stride=st, padding=pd))
        f1 = timeit(lambda: torch.ops.aten.convolution_backward(dyr, xr, wr, None, [st, st], [pd, pd], [1, 1], False, [0, 0], 1, [True, False, False]))
        f2 = timeit(lambda: torch.ops.aten.convolution_backward(dyr, xr, wr, None, [st, st], [pd, pd], [1, 1], False, [0, 0], 1, [False, True, False]))
        res[name] = [f0, f1, f2]
    tot["ours"] += sum(o); tot["cl"] += sum(res["cl"]); tot["nchw"] += sum(res["nchw"])
    fmt3 = lambda v: "%.3f / %.3f / %.3f" % tuple(v)
    print("%-38s | %-26s | %-26s | %-26s" % (str((N, H, W, Cin, Cout, k, st, pd)), fmt3(o), fmt3(res["cl"]), fmt3(res["nchw"])))
print("sum over the listed layers: ours %.2f ms, torch channels_last %.2f ms, torch NCHW %.2f ms" % (tot["ours"], tot["cl"], tot["nchw"]))
