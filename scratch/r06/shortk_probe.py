"""Round 6: where the short-K 1x1 layers' time goes -- the stage-1 geometry (M = 32 x 128 x 128 output points) with K and N varied,
each kind of call the graph makes (float A / BatchNorm affine in the loader / A as piece planes; plain / statistics / residual),
against the probe's 5.6 TB/s for the same bytes (scratch/r06/bw_probe.hip).  DSPN_XT=0/1 switches the tile-spanning loop."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
B, H, W = 32, 128, 128
M = B * H * W

def timeit(f, reps=5):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

def planes_of(t):
    am = fn.absmax(t)
    one, zero = torch.ones(t.shape[-1], device=dev), torch.zeros(t.shape[-1], device=dev)
    return fn.bn_apply_planes(t, one, zero, am), am

print("%5s %5s | %-34s %8s %6s" % ("K", "N", "call", "us", "TB/s"))
for K, N in ((64, 256), (64, 128), (32, 256), (128, 256), (256, 256), (256, 128), (256, 64), (128, 512)):
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(B, H, W, K, generator=g).to(dev)
    w = (torch.randn(N, 1, 1, K, generator=g) / K ** 0.5).to(dev)
    res = torch.randn(B, H, W, N, generator=g).to(dev)
    y = torch.empty(B, H, W, N, device=dev)
    aff = ((torch.rand(K, generator=g) + 0.5).to(dev), torch.randn(K, generator=g).to(dev), True)
    xa, xaa = fn.absmax(x), fn.absmax(x, aff)
    xp, xpa = planes_of(x.abs())
    wa = fn.absmax(w); wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    t2, _ = fn.conv_stats_layout(M, N)
    st = torch.zeros(t2, 2, N, device=dev); mm = torch.zeros(t2, 2, N, device=dev)
    base = dict(w_planes=wp, w_absmax=wa, out=y)
    byt = M * K * 4 + M * N * 4
    calls = [
        ("float A, plain", byt, lambda: fn.conv2d_forward(x, w, None, 1, 0, 1, x_absmax=xa, **base)),
        ("float A, statistics", byt, lambda: fn.conv2d_forward(x, w, None, 1, 0, 1, x_absmax=xa, out_stats=st, out_minmax=mm, **base)),
        ("float A + affine, plain", byt, lambda: fn.conv2d_forward(x, w, None, 1, 0, 1, x_absmax=xaa, in_affine=aff, **base)),
        ("float A + affine, statistics", byt, lambda: fn.conv2d_forward(x, w, None, 1, 0, 1, x_absmax=xaa, in_affine=aff, out_stats=st, out_minmax=mm, **base)),
        ("float A + affine, residual", byt + M * N * 4, lambda: fn.conv2d_forward(x, w, None, 1, 0, 1, x_absmax=xaa, in_affine=aff, residual=res, **base)),
        ("planes A, plain", byt, lambda: fn.conv2d_forward(xp, w, None, 1, 0, 1, x_absmax=xpa, x_planes=True, **base)),
        ("planes A, statistics", byt, lambda: fn.conv2d_forward(xp, w, None, 1, 0, 1, x_absmax=xpa, x_planes=True, out_stats=st, out_minmax=mm, **base)),
    ]
    for name, nb, f in calls:
        t = timeit(f)
        print("%5d %5d | %-34s %8.1f %6.2f" % (K, N, name, t, nb / t / 1e6))
