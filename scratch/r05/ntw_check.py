"""Round 5: the wide tile family (conv_wide.h) against conv_nt_kernel on plane-fed layers: identical output bits, BatchNorm
tables within rounding, and the time of every tile shape.  python scratch/r05/ntw_check.py [quick]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dspnet_amd import functional as fn
from dspnet_amd import _lib
L = _lib.lib()
NAMES = {1: "narrow", 2: "256x128", 3: "128x256", 4: "128x128w"}

def timeit(f, reps=20):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

def planes_of(t):
    am = fn.absmax(t)
    one, zero = torch.ones(t.shape[-1], device="cuda"), torch.zeros(t.shape[-1], device="cuda")
    return fn.bn_apply_planes(t, one, zero, am), am

def forward_case(N, H, W, Cin, Cout, k, stride, quick):
    g = torch.Generator().manual_seed(H + Cin + Cout + k)
    pad = k // 2
    x = torch.randn(N, H, W, Cin, generator=g).cuda().abs_()       # (a ReLU output)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    res = torch.randn(N, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1, Cout, generator=g).cuda()
    xp, xa = planes_of(x)
    wa = fn.absmax(w); wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    Ho = (H + 2 * pad - k) // stride + 1; Wo = (W + 2 * pad - k) // stride + 1
    t2, rows = fn.conv_stats_layout(N * Ho * Wo, Cout)
    out = {}
    for mode in (1, 2, 3, 4):
        if mode == 3 and Cout % 256: continue
        if mode in (2, 4) and Cout % 128: continue
        L.dspn_conv_set_wide_tiles(mode)
        st = torch.zeros(t2, 2, Cout, device="cuda"); mm = torch.zeros(t2, 2, Cout, device="cuda")
        y = fn.conv2d_forward(xp, w, None, stride, pad, 1, out_stats=st, out_minmax=mm, w_planes=wp, x_absmax=xa, w_absmax=wa, x_planes=True)
        y2 = fn.conv2d_forward(xp, w, None, stride, pad, 1, w_planes=wp, x_absmax=xa, w_absmax=wa, x_planes=True, residual=res, relu=True)
        torch.cuda.synchronize()
        tt = None
        if not quick:
            yy = torch.empty_like(y)
            tt = (timeit(lambda: fn.conv2d_forward(xp, w, None, stride, pad, 1, out=yy, w_planes=wp, x_absmax=xa, w_absmax=wa, x_planes=True)),
                  timeit(lambda: fn.conv2d_forward(xp, w, None, stride, pad, 1, out=yy, out_stats=st, out_minmax=mm, w_planes=wp, x_absmax=xa, w_absmax=wa, x_planes=True)))
        out[mode] = (y, st, mm, y2, tt)
    ref = out[1]
    flop = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    line = "fwd  %-28s" % str((N, H, W, Cin, Cout, k, stride))
    for mode, (y, st, mm, y2, tt) in out.items():
        ok = torch.equal(y, ref[0]) and torch.equal(y2, ref[3]) and torch.equal(mm, ref[2])
        se = float((st - ref[1]).abs().max() / (ref[1].abs().max() + 1e-30))
        ok = ok and se < 1e-5
        line += " | %s %s" % (NAMES[mode], "ok" if ok else "MISMATCH(st %.1e, y %.1e)" % (se, float((y - ref[0]).abs().max())))
        if tt: line += " %.0f/%.0f us %.0f TF" % (tt[0], tt[1], flop / tt[0] * 1e-6)
    print(line, flush=True)

def dgrad_case(N, H, W, Cin, Cout, k, stride, quick):
    """data gradient of a (Cin -> Cout, k x k, stride) convolution at input size H x W with dy as planes, + BatchNorm-backward sums"""
    g = torch.Generator().manual_seed(H + Cin + Cout + k + 7)
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1; Wo = (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    dyp, dya = planes_of(dy)
    wa = fn.absmax(w); wt = fn.weight_transpose(w)
    wtp = fn.weight_planes(w, transposed=True, cols=Cout, math="f16x2", w_absmax=wa)
    gamma = torch.rand(Cin, device="cuda") + 0.5; beta = torch.randn(Cin, device="cuda")
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    tiles = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
    out = {}
    for mode in (1, 2, 3, 4):
        if mode == 3 and Cin % 256: continue
        if mode in (2, 4) and Cin % 128: continue
        L.dspn_conv_set_wide_tiles(mode)
        sums = torch.zeros(tiles, 2, Cin, device="cuda"); bam = torch.zeros(64, device="cuda")
        dx = torch.empty_like(x); dx2 = torch.empty_like(x)
        fn.conv2d_dgrad(dyp, wt, tuple(x.shape), stride, pad, 1, out=dx, wt_planes=wtp, dy_absmax=dya, w_absmax=wa,
                        bn_bwd=(x, scale, shift, mean, rstd, True, sums), bn_dy_absmax=bam, dy_planes=True)
        fn.conv2d_dgrad(dyp, wt, tuple(x.shape), stride, pad, 1, out=dx2, wt_planes=wtp, dy_absmax=dya, w_absmax=wa, dy_planes=True)
        torch.cuda.synchronize()
        tt = None
        if not quick:
            tt = (timeit(lambda: fn.conv2d_dgrad(dyp, wt, tuple(x.shape), stride, pad, 1, out=dx2, wt_planes=wtp, dy_absmax=dya, w_absmax=wa, dy_planes=True)),
                  timeit(lambda: fn.conv2d_dgrad(dyp, wt, tuple(x.shape), stride, pad, 1, out=dx, wt_planes=wtp, dy_absmax=dya, w_absmax=wa,
                                                 bn_bwd=(x, scale, shift, mean, rstd, True, sums), bn_dy_absmax=bam, dy_planes=True)))
        out[mode] = (dx, sums, bam.max().clone(), dx2, tt)
    ref = out[1]
    flop = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    line = "dgrd %-28s" % str((N, H, W, Cin, Cout, k, stride))
    for mode, (dx, sums, bam, dx2, tt) in out.items():
        se = float((sums - ref[1]).abs().max() / (ref[1].abs().max() + 1e-30))
        ok = torch.equal(dx, ref[0]) and torch.equal(dx2, ref[3]) and float(bam) == float(ref[2]) and se < 2e-5
        line += " | %s %s" % (NAMES[mode], "ok" if ok else "MISMATCH(sums %.1e, dx %.1e, am %g %g)" % (se, float((dx - ref[0]).abs().max()), float(bam), float(ref[2])))
        if tt: line += " %.0f/%.0f us %.0f TF" % (tt[0], tt[1], flop / tt[0] * 1e-6)
    print(line, flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "shortk":
    B = 32
    for c in [(B, 128, 128, 64, 256, 1, 1), (B, 128, 128, 64, 128, 1, 1), (B, 64, 64, 128, 512, 1, 1), (B, 64, 64, 256, 512, 1, 1), (B, 32, 32, 256, 1024, 1, 1)]:
        forward_case(*c, False)
    for c in [(B, 128, 128, 256, 64, 1, 1), (B, 128, 128, 128, 64, 1, 1), (B, 64, 64, 512, 128, 1, 1), (B, 64, 64, 256, 128, 1, 1)]:
        dgrad_case(*c, False)
    L.dspn_conv_set_wide_tiles(0)
    sys.exit(0)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
small = [(2, 24, 24, 64, 128, 3, 1), (3, 25, 23, 96, 256, 3, 2), (1, 17, 19, 128, 256, 1, 1), (2, 9, 9, 256, 512, 3, 1), (1, 40, 40, 32, 384, 3, 1)]
for c in small:
    forward_case(*c, True)
for c in [(2, 24, 24, 128, 64, 3, 1), (3, 25, 23, 256, 96, 3, 2), (1, 17, 19, 256, 128, 1, 1), (2, 9, 9, 512, 256, 3, 1), (2, 16, 16, 128, 64, 1, 2)]:
    dgrad_case(*c, True)
if not quick:
    B = 32
    for c in [(B, 64, 64, 128, 128, 3, 1), (B, 32, 32, 256, 256, 3, 1), (B, 16, 16, 512, 512, 3, 1), (B, 64, 64, 256, 256, 3, 2),
              (B, 32, 32, 256, 1024, 1, 1), (B, 16, 16, 512, 2048, 1, 1), (B, 32, 32, 1024, 256, 1, 1), (B, 16, 16, 2048, 512, 1, 1), (B, 64, 64, 512, 128, 1, 1)]:
        forward_case(*c, False)
    for c in [(B, 64, 64, 128, 128, 3, 1), (B, 32, 32, 256, 256, 3, 1), (B, 16, 16, 512, 512, 3, 1), (B, 64, 64, 256, 256, 3, 2),
              (B, 32, 32, 1024, 256, 1, 1), (B, 16, 16, 2048, 512, 1, 1), (B, 64, 64, 512, 128, 1, 1)]:
        dgrad_case(*c, False)
L.dspn_conv_set_wide_tiles(0)
