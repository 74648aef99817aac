"""what would conv0 cost as a 7x1 convolution over a row-expanded input (8 horizontal taps x 4 channels = 32 channels per
output column, as fp16 piece planes)?  Same M, N, K' = 224 and output bytes as the real layer; the vertical stride is 1 here
(256 input rows instead of 512)."""
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
N = 32
# today's layer
x = torch.randn(N, 512, 512, 4, device="cuda"); w = torch.randn(64, 7, 7, 4, device="cuda") * 0.05
tiles, _ = fn.conv_stats_layout(N * 256 * 256, 64)
st = torch.zeros(tiles, 2, 64, device="cuda"); mm = torch.zeros(tiles, 2, 64, device="cuda")
y = torch.empty(N, 256, 256, 64, device="cuda"); dy = torch.randn(N, 256, 256, 64, device="cuda")
ax, aw, ady = fn.absmax(x), fn.absmax(w), fn.absmax(dy)
print("conv0 today: fwd %.1f us  wgrad %.1f us" % (
    timeit(lambda: fn.conv2d_forward(x, w, None, 2, 3, 1, out=y, out_stats=st, out_minmax=mm, x_absmax=ax, w_absmax=aw)),
    timeit(lambda: fn.conv2d_wgrad(x, dy, tuple(w.shape), 2, 3, 1, x_absmax=ax, dy_absmax=ady))))
# row-expanded stand-in
xr = torch.randn(N, 256, 256, 32, device="cuda"); wr = torch.randn(64, 7, 1, 32, device="cuda") * 0.05
axr, awr = fn.absmax(xr), fn.absmax(wr)
one, zero = torch.ones(32, device="cuda"), torch.zeros(32, device="cuda")
pl = fn.bn_apply_planes(xr, one, zero, axr)
wp = fn.weight_planes(wr, math="f16x2", w_absmax=awr)
print("7x1 over 32-channel planes: fwd %.1f us  wgrad %.1f us  (floats: fwd %.1f us)  planes write %.1f us" % (
    timeit(lambda: fn.conv2d_forward(pl, wr, None, 1, (3, 0), 1, out=y, out_stats=st, out_minmax=mm, x_absmax=axr, w_absmax=awr, w_planes=wp, x_planes=True)),
    timeit(lambda: fn.conv2d_wgrad(pl, dy, tuple(wr.shape), 1, (3, 0), 1, x_absmax=axr, dy_absmax=ady, x_planes=True)),
    timeit(lambda: fn.conv2d_forward(xr, wr, None, 1, (3, 0), 1, out=y, out_stats=st, out_minmax=mm, x_absmax=axr, w_absmax=awr, w_planes=wp)),
    timeit(lambda: fn.bn_apply_planes(xr, one, zero, axr, out=pl))))
