"""Round 5: conv_ntv_kernel (float A operand; rows requested two k-steps ahead by asm loads, pipelined epilogue reads) against
conv_nt_kernel on RANDOM shapes: identical output bits, identical extremes, statistics within rounding.
python scratch/r05/ntv_random_check.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dspnet_amd import functional as fn, _lib
L = _lib.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for it in range(cases):
    k = int(rng.choice([1, 1, 3, 3, 5]))
    stride = int(rng.choice([1, 1, 2]))
    Cin = 32 * int(rng.integers(1, 9))
    Cout = int(rng.choice([32, 48, 64, 96, 128, 160, 256, 384, 512]))
    N = int(rng.integers(1, 5))
    H, W = int(rng.integers(7, 70)), int(rng.integers(7, 70))
    pad = int(rng.choice([0, k // 2]))
    if (H + 2 * pad - k) < 0 or (W + 2 * pad - k) < 0:
        continue
    affine = bool(rng.integers(0, 2)); relu = bool(rng.integers(0, 2)); use_res = bool(rng.integers(0, 2)); stats = bool(rng.integers(0, 2))
    g = torch.Generator().manual_seed(it)
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, Ho, Wo, Cout, generator=g).cuda() if (use_res and not stats) else None
    ia = (torch.rand(Cin, generator=g).cuda() + 0.5, torch.randn(Cin, generator=g).cuda(), True) if affine else None
    xa, wa = fn.absmax(x, ia), fn.absmax(w)
    wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    outs = {}
    for mode in (1, 0):
        L.dspn_conv_set_wide_tiles(mode)
        st = mm = None
        if stats and Cout % 4 == 0:
            t2, rows = fn.conv_stats_layout(N * Ho * Wo, Cout)
            if t2:
                st = torch.zeros(t2, 2, Cout, device="cuda"); mm = torch.zeros(t2, 2, Cout, device="cuda")
        y = fn.conv2d_forward(x, w, None, stride, pad, 1, relu=relu and st is None, residual=res, in_affine=ia, out_stats=st,
                              out_minmax=mm, w_planes=wp, x_absmax=xa, w_absmax=wa, math="f16x2")
        torch.cuda.synchronize()
        outs[mode] = (y, st, mm)
    L.dspn_conv_set_wide_tiles(0)
    (y1, s1, m1), (y0, s0, m0) = outs[1], outs[0]
    ok = torch.equal(y1, y0) and (m1 is None or torch.equal(m1, m0))
    if s1 is not None:
        ok = ok and float((s1 - s0).abs().max() / (s1.abs().max() + 1e-30)) < 1e-5
    if not ok:
        bad += 1
    print("%s N %d %dx%d Cin %d Cout %d k %d s %d pad %d affine %d relu %d res %d stats %d" % (
        "ok      " if ok else "MISMATCH", N, H, W, Cin, Cout, k, stride, pad, affine, relu, res is not None, s1 is not None), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
