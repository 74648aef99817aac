"""Round 6: the plane-fed weight gradient on the wide tile (conv_wgw_kernel) against conv_wgrad_kernel, layer by layer at the
bench shapes (batch 32, 512 x 512 input): alternating launches on one box, HIP events.  usage: python scratch/r06/wgw_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dspnet_amd import _lib, functional as fn

L = _lib.lib()
# (name, N, H, W, Cin, Cout, k, stride)
LAYERS = [("stage1 conv2", 32, 128, 128, 64, 64, 3, 1), ("stage2 unit1 conv2", 32, 128, 128, 128, 128, 3, 2),
          ("stage2 conv2", 32, 64, 64, 128, 128, 3, 1), ("stage3 unit1 conv2", 32, 64, 64, 256, 256, 3, 2),
          ("stage3 conv2", 32, 32, 32, 256, 256, 3, 1), ("stage4 unit1 conv2", 32, 32, 32, 512, 512, 3, 2),
          ("stage4 conv2", 32, 16, 16, 512, 512, 3, 1), ("stage3 unit1 conv1", 32, 64, 64, 512, 256, 1, 1),
          ("stage4 unit1 conv1", 32, 32, 32, 1024, 512, 1, 1)]
# the layers with one or two FLOAT operands: (name, ..., x planes, dy planes, affine on x)
MIXED = [("stage2 conv1 (x float+aff, dy pl)", 32, 64, 64, 512, 128, 1, 1, False, True, True),
         ("stage3 conv1 (x float+aff, dy pl)", 32, 32, 32, 1024, 256, 1, 1, False, True, True),
         ("stage4 conv1 (x float+aff, dy pl)", 32, 16, 16, 2048, 512, 1, 1, False, True, True),
         ("stage2 conv3 (x pl, dy float)", 32, 64, 64, 128, 512, 1, 1, True, False, False),
         ("stage3 conv3 (x pl, dy float)", 32, 32, 32, 256, 1024, 1, 1, True, False, False),
         ("stage4 conv3 (x pl, dy float)", 32, 16, 16, 512, 2048, 1, 1, True, False, False),
         ("stage1 conv3 (both float+aff)", 32, 128, 128, 64, 256, 1, 1, False, False, True),
         ("stage2 sc (both float+aff)", 32, 128, 128, 256, 512, 1, 2, False, False, True),
         ("stage3 sc (x pl, dy float)", 32, 64, 64, 512, 1024, 1, 2, True, False, False)]


def planes_of(t):
    am = fn.absmax(t)
    one, zero = torch.ones(t.shape[-1], device="cuda"), torch.zeros(t.shape[-1], device="cuda")
    return fn.bn_apply_planes(t, one, zero, am), am


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("%-36s %9s %9s %7s   TF old / new (of 833.3)" % ("layer", "old us", "wide us", "ratio"))
tot = [0.0, 0.0]
for row in [l + (True, True, False) for l in LAYERS] + MIXED:
    name, N, H, W, Cin, Cout, k, stride, x_pl, dy_pl, affine = row
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, device="cuda").abs_()
    dy = torch.randn(N, Ho, Wo, Cout, device="cuda")
    aff = ((torch.rand(Cin, device="cuda") + 0.5), torch.randn(Cin, device="cuda"), True) if affine else None
    if affine:
        x = torch.randn(N, H, W, Cin, device="cuda")
    xa = fn.absmax(x, aff) if affine else fn.absmax(x)
    dya = fn.absmax(dy)
    xp = planes_of(x)[0] if x_pl else x
    dyp = planes_of(dy)[0] if dy_pl else dy
    del x, dy
    wshape = (Cout, k, k, Cin)
    splits = fn.conv2d_wgrad_splits((N, H, W, Cin), (N, Ho, Wo, Cout), wshape, stride)
    slabs = torch.zeros(max(splits, 1), Cout * k * k * Cin, device="cuda")
    f = lambda: fn.conv2d_wgrad_slabs(xp, dyp, wshape, slabs, stride, pad, 1, x_absmax=xa, dy_absmax=dya, x_planes=x_pl, dy_planes=dy_pl, in_affine=aff)
    ts = []
    for rep in range(2):
        for mode in (1, 0):
            L.dspn_conv_set_wide_tiles(mode)
            ts.append(timeit(f))
    L.dspn_conv_set_wide_tiles(0)
    old, new = min(ts[0], ts[2]), min(ts[1], ts[3])
    fl = 2.0 * N * Ho * Wo * Cout * k * k * Cin
    tot[0] += old; tot[1] += new
    print("%-36s %9.1f %9.1f %7.3f   %.0f / %.0f  (%.2f / %.2f)  splits %d" % (name, old, new, new / old, fl / old / 1e6, fl / new / 1e6,
                                                                             fl / old / 1e6 / 833.3, fl / new / 1e6 / 833.3, splits))
print("sum %.1f -> %.1f us" % tuple(tot))
