"""data gradient of one layer with dy as floats / as fp16 piece planes (timing only), with and without the BatchNorm-backward epilogue"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspnet_amd import functional as fn
def timeit(f, reps=20):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (N, H, W, Cin, Cout, k) in [(32, 128, 128, 256, 64, 1), (32, 64, 64, 512, 128, 1), (32, 32, 32, 1024, 256, 1), (32, 16, 16, 2048, 512, 1),
                                (32, 64, 64, 128, 128, 3), (32, 32, 32, 256, 256, 3)]:
    x = torch.randn(N, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    dy = torch.randn(N, H, W, Cout, device="cuda").abs() * 1e-3 + 1e-3          # (as planes: finite fp16 bit patterns)
    dx = torch.empty_like(x)
    wa = fn.absmax(w); wt = fn.weight_transpose(w)
    wtp = fn.weight_planes(w, transposed=True, cols=Cout, math="f16x2", w_absmax=wa)
    dya = fn.absmax(dy)
    gamma = torch.rand(Cin, device="cuda") + 0.5; beta = torch.randn(Cin, device="cuda")
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    tiles = fn.conv_dgrad_bn_tiles(tuple(x.shape), 1)
    sums = torch.zeros(tiles, 2, Cin, device="cuda")
    bn = (x, scale, shift, mean, rstd, True, sums)
    r = []
    for planes in (False, True):
        for b in (None, bn):
            r.append(timeit(lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), 1, k // 2, 1, out=dx, wt_planes=wtp, dy_absmax=dya, w_absmax=wa,
                                                    bn_bwd=b, dy_planes=planes)))
    print((N, H, W, Cin, Cout, k), "floats: plain %.1f us, +bn sums %.1f us | planes: plain %.1f us, +bn sums %.1f us" % tuple(r))
