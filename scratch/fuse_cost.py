"""cost of the fused BatchNorm pieces per forward convolution: plain | +input affine | +output statistics | both"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
DT = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32   # tensor storage type
def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
shapes = [(32, 16, 16, 512, 512, 3), (32, 32, 32, 256, 256, 3), (32, 64, 64, 128, 128, 3), (32, 128, 128, 64, 64, 3),
          (32, 32, 32, 1024, 256, 1), (32, 32, 32, 256, 1024, 1), (32, 64, 64, 128, 512, 1), (32, 128, 128, 256, 64, 1), (32, 128, 128, 64, 256, 1)]
tot = [0, 0, 0, 0]
for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev).to(DT); w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.05).to(DT)
    o = torch.empty(N, H, W, Cout, device=dev, dtype=DT)
    aff = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev), True)
    tiles, _ = fn.conv_stats_layout(N * H * W, Cout)
    st = torch.empty(tiles, 2, Cout, device=dev)
    kw, kwa, mm = {}, {}, {}
    if DT == torch.float32 and fn.get_conv_math() == "f16x2":     # magnitudes / planes / min-max table as the graph passes them
        wa = fn.absmax(w)
        wp = fn.weight_planes(w, math="f16x2", w_absmax=wa) if Cin % 32 == 0 else None
        kw = dict(x_absmax=fn.absmax(x), w_absmax=wa, w_planes=wp)
        kwa = dict(x_absmax=fn.absmax(x, aff), w_absmax=wa, w_planes=wp)
        mm = dict(out_minmax=torch.empty(tiles, 2, Cout, device=dev))
    t = [timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=o, **kw)),
         timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=o, in_affine=aff, **kwa)),
         timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=o, out_stats=st, **kw, **mm)),
         timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=o, in_affine=aff, out_stats=st, **kwa, **mm))]
    for i in range(4): tot[i] += t[i]
    print((N, H, W, Cin, Cout, k), "plain %.3f | +affine %.3f (%+.0f%%) | +stats %.3f (%+.0f%%) | both %.3f (%+.0f%%)" % (
        t[0], t[1], 100 * (t[1] / t[0] - 1), t[2], 100 * (t[2] / t[0] - 1), t[3], 100 * (t[3] / t[0] - 1)))
print("totals", ["%.3f" % v for v in tot])
