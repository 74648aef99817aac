"""Round 5: the wide tile family of the two-piece convolution math (csrc/conv_wide.h) -- both operands as fp16 piece planes,
global -> LDS directly, 64 x 64 outputs per wave.  It must give the BITS of conv_nt_kernel (same K order, same accumulation
order per output, same epilogue arithmetic) on every tile shape, forward and data gradient, with every epilogue: plain,
bias / residual / ReLU / accumulate, BatchNorm statistics + extremes, BatchNorm-backward sums + largest stored gradient;
and both must agree with the fp32-MFMA convolution of the decoded operands to fp32 accuracy."""
import numpy as np
import pytest
import torch

from dspnet_amd import _lib
from dspnet_amd import functional as fn

pytestmark = pytest.mark.gpu
MODES = {1: "conv_nt_kernel", 2: "256x128", 3: "128x256", 4: "128x128 on four waves"}


@pytest.fixture()
def tiles(gpu_device):
    if fn.get_conv_math() != "f16x2":
        pytest.skip("the two-piece math is not this process's default")
    L = _lib.lib()
    yield lambda mode: _lib.check(L.dspn_conv_set_wide_tiles(mode), "set_wide_tiles")
    L.dspn_conv_set_wide_tiles(0)


def planes_of(t):
    am = fn.absmax(t)
    one, zero = torch.ones(t.shape[-1], device="cuda"), torch.zeros(t.shape[-1], device="cuda")
    return fn.bn_apply_planes(t, one, zero, am), am


def test_setter_rejects_unknown_modes(gpu_device):
    L = _lib.lib()
    assert L.dspn_conv_set_wide_tiles(5) != 0 and b"conv_set_wide_tiles" in L.dspn_last_error()
    assert L.dspn_conv_set_wide_tiles(-1) != 0
    assert L.dspn_conv_set_wide_tiles(0) == 0


# (N, H, W, Cin, Cout, k, stride): tile counts around one / many per CU, M and Cout not multiples of the tiles, one k-step
FWD = [(2, 24, 24, 64, 128, 3, 1), (3, 25, 23, 96, 256, 3, 2), (1, 17, 19, 128, 256, 1, 1), (2, 9, 9, 256, 512, 3, 1),
       (1, 40, 40, 32, 384, 3, 1), (2, 31, 33, 64, 192, 3, 1), (4, 64, 64, 32, 96, 1, 1), (8, 32, 32, 128, 320, 3, 1),
       # at most 64 output columns: the 256 x 64 tile with 64-row BatchNorm tables (stage 1 of the ResNets)
       (4, 64, 64, 64, 64, 3, 1), (3, 33, 31, 128, 48, 3, 1), (2, 64, 64, 128, 64, 1, 1)]


@pytest.mark.parametrize("case", FWD)
def test_forward_on_every_tile_is_bit_identical(tiles, case):
    N, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(H + Cin + Cout + k)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).cuda().abs_()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    res = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    base = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    xp, xa = planes_of(x)
    wa = fn.absmax(w); wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    t2, rows = fn.conv_stats_layout(N * Ho * Wo, Cout)
    out = {}
    for mode in MODES:
        tiles(mode)
        kw = dict(w_planes=wp, x_absmax=xa, w_absmax=wa, x_planes=True)
        y_plain = fn.conv2d_forward(xp, w, None, stride, pad, 1, **kw)
        got = [y_plain, fn.conv2d_forward(xp, w, bias, stride, pad, 1, residual=res, relu=True, **kw)]
        acc = base.clone()
        fn.conv2d_forward(xp, w, None, stride, pad, 1, out=acc, accumulate=True, **kw)
        got.append(acc)
        if t2 > 0:
            st = torch.zeros(t2, 2, Cout, device="cuda"); mm = torch.zeros(t2, 2, Cout, device="cuda")
            got += [fn.conv2d_forward(xp, w, None, stride, pad, 1, out_stats=st, out_minmax=mm, **kw), mm, st]
        out[mode] = got
    ref32 = fn.conv2d_forward(x, w, None, stride, pad, 1, math="fp32")
    assert float((out[1][0] - ref32).abs().max()) <= 1e-5 * float(ref32.abs().max())
    for mode in (2, 3, 4):
        for i, (a, b) in enumerate(zip(out[mode], out[1])):
            if i == 5:      # per-tile (mean, M2): the row groups of a thread differ between tile shapes -- fp32 rounding only
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (MODES[mode], "statistics")
            else:
                assert torch.equal(a, b), (MODES[mode], i)


# data gradient of a (Cin -> Cout, k x k, stride) convolution at input size H x W; Cin plays the role of the output columns
BWD = [(2, 24, 24, 128, 64, 3, 1), (3, 25, 23, 256, 96, 3, 2), (1, 17, 19, 256, 128, 1, 1), (2, 9, 9, 512, 256, 3, 1),
       (2, 16, 16, 128, 64, 1, 2), (2, 30, 34, 192, 64, 3, 1), (8, 32, 32, 320, 128, 3, 2),
       (4, 64, 64, 64, 64, 3, 1), (3, 33, 31, 48, 128, 3, 1), (2, 64, 64, 64, 256, 1, 1), (2, 64, 64, 64, 128, 3, 2)]


@pytest.mark.parametrize("case", BWD)
def test_data_gradient_on_every_tile_is_bit_identical(tiles, case):
    N, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(H + Cin + Cout + k + 7)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    dyp, dya = planes_of(dy)
    wa = fn.absmax(w); wt = fn.weight_transpose(w)
    wtp = fn.weight_planes(w, transposed=True, cols=Cout, math="f16x2", w_absmax=wa)
    gamma = torch.rand(Cin, device="cuda") + 0.5; beta = torch.randn(Cin, device="cuda")
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    ntile = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
    out = {}
    for mode in MODES:
        tiles(mode)
        kw = dict(wt_planes=wtp, dy_absmax=dya, w_absmax=wa, dy_planes=True)
        sums = torch.zeros(ntile, 2, Cin, device="cuda"); bam = torch.zeros(64, device="cuda")
        dx = torch.empty_like(x); dx2 = torch.empty_like(x); dx3 = x.clone()
        fn.conv2d_dgrad(dyp, wt, tuple(x.shape), stride, pad, 1, out=dx, bn_bwd=(x, scale, shift, mean, rstd, True, sums),
                        bn_dy_absmax=bam, **kw)
        fn.conv2d_dgrad(dyp, wt, tuple(x.shape), stride, pad, 1, out=dx2, **kw)
        fn.conv2d_dgrad(dyp, wt, tuple(x.shape), stride, pad, 1, out=dx3, accumulate=True, **kw)
        out[mode] = [dx, dx2, dx3, bam.max().reshape(1).clone(), sums]
    ref32 = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, math="fp32")
    assert float((out[1][1] - ref32).abs().max()) <= 1e-5 * float(ref32.abs().max())
    assert float(out[1][3]) == float(out[1][0].abs().max()), "bn_dy_absmax is the largest stored gradient"
    for mode in (2, 3, 4):
        for i, (a, b) in enumerate(zip(out[mode], out[1])):
            if i == 4:
                assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (MODES[mode], "BatchNorm-backward sums")
            else:
                assert torch.equal(a, b), (MODES[mode], i)


def test_automatic_choice_is_one_of_the_tested_paths(tiles):
    """mode 0 (what the graph runs): same bits as mode 1 on a layer the automatic policy routes to the wide family"""
    N, H, W, Cin, Cout, k = 8, 32, 32, 256, 256, 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).cuda().abs_()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    xp, xa = planes_of(x)
    wa = fn.absmax(w); wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    ys = []
    for mode in (1, 0):
        tiles(mode)
        ys.append(fn.conv2d_forward(xp, w, None, 1, 1, 1, w_planes=wp, x_absmax=xa, w_absmax=wa, x_planes=True))
    assert torch.equal(ys[0], ys[1])


@pytest.mark.parametrize("mode", [1, 4])
def test_infinite_activation_through_the_planes_gives_fp32s_infinities(tiles, mode):
    """VERDICT r04 item 4 (second half): an infinite ACTIVATION.  The plane writers repair an infinite element to the pieces
    (+-65504, +-inf) (dspn_bn_apply_planes_f32 since round 4, the BatchNorm backward's planes since round 5), so every plane-fed
    call -- conv_nt_kernel's EPIX & 4 path and the wide family alike -- gives x w = h0 g0 + h0 g1 + h1 g0 = +-inf with fp32's
    sign, NaN where the weight is 0 (inf * 0), and leaves every output the element does not touch as accurate as before."""
    N, H, W, Cin, Cout = 2, 12, 12, 64, 128
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) / 24
    inf = float("inf")
    x[0, 3, 3, 1] = inf; x[0, 8, 2, 5] = -inf; x[1, 5, 5, 2] = float("nan"); x[1, 9, 9, 7] = inf
    w[5, 1, 1, 1] = 0.0                      # inf * 0 -> NaN at (0, 3, 3) of output channel 5
    x, w = x.cuda(), w.cuda()
    ref = fn.conv2d_forward(x, w, None, 1, 1, 1, math="fp32")
    xp, xa = planes_of(x)
    wa = fn.absmax(w); wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    tiles(mode)
    got = fn.conv2d_forward(xp, w, None, 1, 1, 1, w_planes=wp, x_absmax=xa, w_absmax=wa, x_planes=True)
    fin = torch.isfinite(ref)
    assert 0 < int((~fin).sum()) < ref.numel() // 2
    assert torch.equal(torch.isfinite(got), fin)
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    assert torch.equal(got[torch.isinf(ref)], ref[torch.isinf(ref)])        # the same signed infinities
    assert float((got[fin] - ref[fin]).abs().max()) <= 1e-5 * float(ref[fin].abs().max())
    # ... and an infinite OUTPUT GRADIENT through the data gradient (dy as planes)
    dy = torch.randn(N, H, W, Cout, generator=g)
    dy[0, 6, 6, 3] = inf; dy[1, 2, 9, 11] = -inf
    dy = dy.cuda()
    w2 = (torch.randn(Cout, 3, 3, Cin, generator=g) / 24).cuda()
    wt = fn.weight_transpose(w2)
    ref = fn.conv2d_dgrad(dy, wt, tuple(x.shape), 1, 1, 1, math="fp32")
    dyp, dya = planes_of(dy)
    w2a = fn.absmax(w2); wtp = fn.weight_planes(w2, transposed=True, cols=Cout, math="f16x2", w_absmax=w2a)
    got = fn.conv2d_dgrad(dyp, wt, tuple(x.shape), 1, 1, 1, wt_planes=wtp, dy_absmax=dya, w_absmax=w2a, dy_planes=True)
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(got), fin) and torch.equal(torch.isnan(got), torch.isnan(ref))
    assert torch.equal(got[torch.isinf(ref)], ref[torch.isinf(ref)])
    assert float((got[fin] - ref[fin]).abs().max()) <= 1e-5 * float(ref[fin].abs().max())


@pytest.mark.parametrize("shape", [(2, 512, 512), (1, 256, 1024), (3, 70, 512)])
def test_stem_convolution_kernel_matches_the_generic_one(tiles, shape):
    """Round 5: the 7x7 / 2 stem convolution (4 physical input channels -> 64, BatchNorm statistics + extremes) on its own
    kernel (csrc/conv_stem.h: a kernel row is one contiguous 32-deep k-step of the image row) against conv_nt_kernel's
    non-uniform-tap path (mode 1 switches the stem kernel off with the wide family) and the fp32 MFMA: outputs to fp32
    accuracy (the two kernels group the 49 taps differently: another summation order), tile statistics / extremes within
    rounding, odd heights and strips that end inside the image."""
    N, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, H, W, 4, generator=g).cuda()
    x[..., 3] = 0                                           # the pad channel of the RGB image
    w = (torch.randn(64, 7, 7, 4, generator=g) / 12).cuda()
    w[..., 3] = 0
    xa, wa = fn.absmax(x), fn.absmax(w)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    t, rows = fn.conv_stats_layout(N * Ho * Wo, 64)
    assert rows == 64
    out = {}
    for mode in (1, 0):
        tiles(mode)
        st = torch.zeros(t, 2, 64, device="cuda"); mm = torch.zeros(t, 2, 64, device="cuda")
        y = fn.conv2d_forward(x, w, None, 2, 3, 1, out_stats=st, out_minmax=mm, x_absmax=xa, w_absmax=wa)
        out[mode] = (y, st, mm)
    ref = fn.conv2d_forward(x, w, None, 2, 3, 1, math="fp32")
    scale = float(ref.abs().max())
    for mode in (1, 0):
        assert float((out[mode][0] - ref).abs().max()) <= 1e-5 * scale, mode
    y, st, mm = out[0]
    # the tables describe the STORED values: recompute them from y
    yt = y.view(-1, 64)[: t * 64].view(t, 64, 64).double() if (N * Ho * Wo) % 64 == 0 else None
    assert yt is not None
    mean = yt.mean(dim=1); m2 = ((yt - mean[:, None, :]) ** 2).sum(dim=1)
    assert float((st[:, 0].double() - mean).abs().max()) <= 1e-5 * scale
    assert float((st[:, 1].double() - m2).abs().max()) <= 1e-4 * float(m2.max())
    assert torch.equal(mm[:, 0], y.view(t, 64, 64).min(dim=1).values) and torch.equal(mm[:, 1], y.view(t, 64, 64).max(dim=1).values)
    assert float((st - out[1][1]).abs().max()) <= 1e-4 * float(out[1][1].abs().max())


def test_stem_shape_without_statistics_still_fills_the_magnitude_block(tiles):
    """Advisor r5: a 7x7 / 2, 4 -> 64 channel convolution called WITHOUT out_stats but WITH out_absmax (an engine stem with no
    BatchNorm behind it) must not take the stem kernel's shortcut, which writes no magnitude block: the block has to hold the
    largest |y| as stored, exactly as the generic path leaves it (a zero block makes the next convolution cut its operand with
    scale 1)."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 128, 512, 4, generator=g).cuda() * 40
    x[..., 3] = 0
    w = (torch.randn(64, 7, 7, 4, generator=g) / 3).cuda()
    w[..., 3] = 0
    xa, wa = fn.absmax(x), fn.absmax(w)
    blocks = {}
    for mode in (1, 0):
        tiles(mode)
        am = torch.zeros(fn.ABSMAX_SLOTS, device="cuda")
        y = fn.conv2d_forward(x, w, None, 2, 3, 1, x_absmax=xa, w_absmax=wa, out_absmax=am)
        assert float(am.max()) == float(y.abs().max()) > 0, mode
        blocks[mode] = (y, am)
    assert torch.equal(blocks[0][0], blocks[1][0])


BF_CASES = [(4, 32, 32, 64, 128, 3, 1), (8, 32, 32, 128, 256, 3, 1), (4, 33, 31, 64, 192, 3, 2), (8, 64, 64, 64, 256, 1, 1),
            (8, 64, 64, 128, 64, 3, 1), (16, 32, 32, 256, 128, 1, 1)]


@pytest.mark.parametrize("case", BF_CASES)
def test_bf16_tensors_on_every_tile_are_bit_identical(gpu_device, case):
    """Round 5: the same family on bfloat16 TENSORS (conv_wide_h.hip): the activations / weight copies are the operands as they
    stand (128-byte records of 64 channels, one MFMA product per block).  Forward with statistics / residual + ReLU, data
    gradient plain / accumulating / with BatchNorm-backward sums: the bits of conv_nt_kernel's bf16 build on every tile."""
    N, H, W, Cin, Cout, k, stride = case
    L = _lib.lib()
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(H + Cin + Cout)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).to(BF).cuda()
    w32 = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).to(BF).float().cuda()
    wh = w32.to(BF)
    wt = fn.weight_transpose(w32, dtype=BF)
    res = torch.randn(N, Ho, Wo, Cout, generator=g).to(BF).cuda()
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(BF).cuda()
    mean = x.float().view(-1, Cin).mean(0)
    rstd = 1.0 / torch.sqrt(x.float().view(-1, Cin).var(0, unbiased=False) + 2e-5)
    scale, shift = rstd.clone(), (-mean * rstd)
    t2, _ = fn.conv_stats_layout(N * Ho * Wo, Cout)
    ntile = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
    out = {}
    try:
        for mode in MODES:
            _lib.check(L.dspn_conv_set_wide_tiles(mode), "set_wide_tiles")
            st = torch.zeros(max(t2, 1), 2, Cout, device="cuda")
            got = [fn.conv2d_forward(x, wh, None, stride, pad, 1), fn.conv2d_forward(x, wh, None, stride, pad, 1, residual=res, relu=True)]
            if t2 > 0:
                got += [fn.conv2d_forward(x, wh, None, stride, pad, 1, out_stats=st), st]
            sums = torch.zeros(ntile, 2, Cin, device="cuda")
            dx = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, bn_bwd=(x, scale, shift, mean, rstd, True, sums))
            acc = x.clone()
            fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=acc, accumulate=True)
            got += [dx, fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1), acc, sums]
            out[mode] = got
    finally:
        L.dspn_conv_set_wide_tiles(0)
    for mode in (2, 3, 4):
        for i, (a, b) in enumerate(zip(out[mode], out[1])):
            if a.dtype == torch.float32:      # statistics / sums: other row groups per thread, fp32 rounding only
                assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6, (MODES[mode], i)
            else:
                assert torch.equal(a, b), (MODES[mode], i)


FLOAT_A = [(2, 24, 24, 64, 128, 3, 1), (3, 25, 23, 96, 256, 3, 2), (1, 17, 19, 128, 256, 1, 1), (4, 64, 64, 64, 64, 3, 1),
           (2, 64, 64, 256, 64, 1, 1), (8, 32, 32, 128, 320, 1, 1), (2, 33, 35, 32, 192, 3, 1)]


@pytest.mark.parametrize("case", FLOAT_A)
@pytest.mark.parametrize("affine", [None, "relu", "linear"])
def test_float_operand_forward_on_every_tile_is_bit_identical(tiles, case, affine):
    """The register-staged member of the family (conv_ntv_kernel): the A operand is a FLOAT tensor, cut into its two pieces in
    the loader, optionally behind the folded BatchNorm affine (+ ReLU; zero padding AFTER the affine).  Same pieces, same K
    order, same epilogue as conv_nt_kernel's INTF / float paths: identical bits on every tile, with every epilogue."""
    N, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(H + Cin + Cout + k + 3)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    res = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    aff = None
    if affine is not None:
        aff = ((torch.rand(Cin, generator=g) + 0.5).cuda(), torch.randn(Cin, generator=g).cuda(), affine == "relu")
    xa = fn.absmax(x, aff)
    wa = fn.absmax(w); wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    t2, _ = fn.conv_stats_layout(N * Ho * Wo, Cout)
    out = {}
    for mode in MODES:
        tiles(mode)
        kw = dict(w_planes=wp, x_absmax=xa, w_absmax=wa, in_affine=aff)
        got = [fn.conv2d_forward(x, w, None, stride, pad, 1, **kw), fn.conv2d_forward(x, w, None, stride, pad, 1, residual=res, relu=True, **kw)]
        if t2 > 0:
            st = torch.zeros(t2, 2, Cout, device="cuda"); mm = torch.zeros(t2, 2, Cout, device="cuda")
            got += [fn.conv2d_forward(x, w, None, stride, pad, 1, out_stats=st, out_minmax=mm, **kw), mm, st]
        out[mode] = got
    act = x if aff is None else fn.bn_apply(x, aff[0], aff[1], relu=aff[2])
    ref32 = fn.conv2d_forward(act, w, None, stride, pad, 1, math="fp32")
    assert float((out[1][0] - ref32).abs().max()) <= 1e-5 * float(ref32.abs().max())
    for mode in (2, 3, 4):
        for i, (a, b) in enumerate(zip(out[mode], out[1])):
            if i == 4:
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (MODES[mode], "statistics")
            else:
                assert torch.equal(a, b), (MODES[mode], i)


@pytest.mark.parametrize("case", [(2, 24, 24, 128, 64, 3, 1), (3, 25, 23, 256, 96, 3, 2), (1, 17, 19, 256, 128, 1, 1),
                                  (4, 64, 64, 64, 256, 1, 1), (2, 16, 16, 128, 64, 1, 2), (8, 32, 32, 320, 128, 1, 1)])
def test_float_gradient_data_gradient_on_every_tile_is_bit_identical(tiles, case):
    """... and the data gradient whose output gradient is a float tensor (the residual stream's: no BatchNorm writes it as
    planes), plain / accumulating / with the BatchNorm-backward sums"""
    N, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(H + Cin + Cout + k + 9)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    dya = fn.absmax(dy)
    wa = fn.absmax(w); wt = fn.weight_transpose(w)
    wtp = fn.weight_planes(w, transposed=True, cols=Cout, math="f16x2", w_absmax=wa)
    gamma = torch.rand(Cin, device="cuda") + 0.5; beta = torch.randn(Cin, device="cuda")
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    ntile = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
    out = {}
    for mode in MODES:
        tiles(mode)
        kw = dict(wt_planes=wtp, dy_absmax=dya, w_absmax=wa)
        sums = torch.zeros(ntile, 2, Cin, device="cuda"); bam = torch.zeros(64, device="cuda")
        dx = torch.empty_like(x); dx2 = torch.empty_like(x); dx3 = x.clone()
        fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=dx, bn_bwd=(x, scale, shift, mean, rstd, True, sums),
                        bn_dy_absmax=bam, **kw)
        fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=dx2, **kw)
        fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=dx3, accumulate=True, **kw)
        out[mode] = [dx, dx2, dx3, bam.max().reshape(1).clone(), sums]
    for mode in (2, 3, 4):
        for i, (a, b) in enumerate(zip(out[mode], out[1])):
            if i == 4:
                assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (MODES[mode], "BatchNorm-backward sums")
            else:
                assert torch.equal(a, b), (MODES[mode], i)


# ---- round 6: the tile-spanning loop (conv_wide.h, XT) of the 128-row members (128 x 128 on four waves, two ring slots, exchange
# behind the ring; 128 x 256 on eight waves, three slots, exchange inside the ring), against their round-5 loop
# (N, H, W, Cin, Cout, k, stride): k-step counts 2 .. 18, one to four tiles per workgroup (512 workgroups fill the chip),
# ragged last row tile / column tile
XT_CASES = [(16, 64, 64, 64, 256, 1, 1), (32, 64, 64, 256, 128, 1, 1), (13, 61, 67, 64, 192, 1, 1), (16, 64, 64, 128, 256, 1, 2),
            (8, 48, 48, 64, 128, 3, 1), (32, 32, 32, 512, 256, 1, 1), (3, 17, 19, 128, 256, 1, 1),
            # odd k-step counts (3 and 9): the ring's slot runs on across tiles, no parity condition
            (16, 64, 64, 96, 256, 1, 1), (8, 64, 64, 32, 128, 3, 1),
            # at most 64 output columns: the 256 x 64 members with 64-row statistics tiles (stage 1 of the ResNets)
            (16, 64, 64, 64, 64, 3, 1), (32, 64, 64, 256, 64, 1, 1), (16, 64, 64, 64, 64, 1, 1)]


@pytest.fixture(params=[4, 3], ids=["128x128 on four waves", "128x256 on eight waves"])
def spanning(tiles, request):
    L = _lib.lib()
    tiles(request.param)
    yield lambda on: _lib.check(L.dspn_conv_set_tile_spanning(on), "set_tile_spanning")
    L.dspn_conv_set_tile_spanning(1)


def test_tile_spanning_setter(gpu_device):
    L = _lib.lib()
    assert L.dspn_conv_set_tile_spanning(3) != 0 and b"conv_set_tile_spanning" in L.dspn_last_error()
    assert L.dspn_conv_set_tile_spanning(1) == 0


@pytest.mark.parametrize("case", XT_CASES)
def test_tile_spanning_loop_gives_the_bits_of_the_plain_loop(spanning, case):
    """VERDICT r05 item 1.  Every kind of call the two kernels serve -- A operand as piece planes (conv_ntw_kernel) or as a
    float tensor with / without the folded BatchNorm affine (conv_ntv_kernel); forward plain, + bias + residual + ReLU,
    accumulating, with statistics and extremes; data gradient plain, accumulating, with the BatchNorm-backward sums and the
    magnitude block -- with the loop that requests the next tile's operands before the current tile's epilogue and with the
    round-5 loop: every stored tensor, the per-tile extremes and the magnitude block bit for bit (same K order, same epilogue
    arithmetic per element); the per-tile BatchNorm tables within rounding (the direct epilogue sums a column in another order)."""
    N, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(H + Cin + Cout + k + 11)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    res = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    base = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    aff = ((torch.rand(Cin, generator=g) + 0.5).cuda(), torch.randn(Cin, generator=g).cuda(), True)
    xpos = x.abs()
    xp, xpa = planes_of(xpos)
    xa, xaa = fn.absmax(x), fn.absmax(x, aff)
    wa = fn.absmax(w); wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    t2, _ = fn.conv_stats_layout(N * Ho * Wo, Cout)
    assert t2 > 0
    # the data gradient of a (Cout -> Cin) convolution with the same geometry: dy has the shape of the forward output
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    dyp, dya = planes_of(dy)
    wt = fn.weight_transpose(w)
    wtp = fn.weight_planes(w, transposed=True, cols=Cout, math="f16x2", w_absmax=wa)
    gamma = torch.rand(Cin, device="cuda") + 0.5; beta = torch.randn(Cin, device="cuda")
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    ntile = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
    out = {}
    for on in (0, 2):          # 2: the plane-fed AND the float-operand kernel on the tile-spanning loop
        spanning(on)
        got, extra = [], []
        for src, kw in ((xp, dict(x_absmax=xpa, x_planes=True)), (x, dict(x_absmax=xa)), (x, dict(x_absmax=xaa, in_affine=aff))):
            kw = dict(kw, w_planes=wp, w_absmax=wa)
            got.append(fn.conv2d_forward(src, w, None, stride, pad, 1, **kw))
            got.append(fn.conv2d_forward(src, w, bias, stride, pad, 1, residual=res, relu=True, **kw))
            acc = base.clone()
            fn.conv2d_forward(src, w, None, stride, pad, 1, out=acc, accumulate=True, **kw)
            st = torch.zeros(t2, 2, Cout, device="cuda"); mm = torch.zeros(t2, 2, Cout, device="cuda")
            got += [acc, fn.conv2d_forward(src, w, None, stride, pad, 1, out_stats=st, out_minmax=mm, **kw), st, mm]
            both = base.clone()          # residual AND accumulate in one call, with bias + ReLU and the magnitude block
            am = torch.zeros(fn.ABSMAX_SLOTS, device="cuda")
            fn.conv2d_forward(src, w, bias, stride, pad, 1, residual=res, relu=True, out=both, accumulate=True, **kw)
            plain_am = fn.conv2d_forward(src, w, None, stride, pad, 1, out_absmax=am, **kw)
            extra.append([both, plain_am, am.max().reshape(1)])
        for src, kw in ((dyp, dict(dy_absmax=dya, dy_planes=True)), (dy, dict(dy_absmax=dya))):
            kw = dict(kw, wt_planes=wtp, w_absmax=wa)
            sums = torch.zeros(ntile, 2, Cin, device="cuda"); bam = torch.zeros(64, device="cuda")
            dx = torch.empty_like(x); dx2 = torch.empty_like(x); dx3 = x.clone()
            fn.conv2d_dgrad(src, wt, tuple(x.shape), stride, pad, 1, out=dx, bn_bwd=(x, scale, shift, mean, rstd, True, sums),
                            bn_dy_absmax=bam, **kw)
            fn.conv2d_dgrad(src, wt, tuple(x.shape), stride, pad, 1, out=dx2, **kw)
            fn.conv2d_dgrad(src, wt, tuple(x.shape), stride, pad, 1, out=dx3, accumulate=True, **kw)
            got += [dx, dx2, dx3, sums, bam.clone()]
        out[on] = got
        out[(on, "extra")] = [t for e in extra for t in e]
    ref32 = fn.conv2d_forward(x, w, None, stride, pad, 1, math="fp32")
    assert float((out[2][6] - ref32).abs().max()) <= 1e-5 * float(ref32.abs().max())
    for i, (a, b) in enumerate(zip(out[2], out[0])):
        if i in (4, 10, 16):        # per-tile (mean, M2): the direct epilogue sums a column's rows in another order -- fp32 rounding only
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (i, "statistics")
        elif i in (21, 26):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (i, "BatchNorm-backward sums")
        else:                       # every stored tensor, the extremes and the magnitude block: the same bits
            assert torch.equal(a, b), i
    for i, (a, b) in enumerate(zip(out[(2, "extra")], out[(0, "extra")])):
        assert torch.equal(a, b), ("extra", i)
    assert float(out[(2, "extra")][2]) == float(out[(2, "extra")][1].abs().max())


# ---- round 6: the weight gradient of the plane-fed layers on the same footing (csrc/conv_wgrad_wide.h) ------------------------
# (N, H, W, Cin, Cout, k, stride, pad, dil): Cout not a multiple of the tile (blocks past Cout are not fetched), J = taps x Cin with
# a ragged last column tile, pixels not a multiple of the k-step, maps smaller than one k-step, stride 2, dilation, many splits
WG_CASES = [(2, 24, 24, 64, 128, 3, 1, 1, 1), (3, 25, 23, 96, 160, 3, 2, 1, 1), (1, 17, 19, 128, 256, 1, 1, 0, 1),
            (5, 5, 5, 256, 192, 3, 1, 1, 1), (2, 20, 20, 32, 96, 5, 1, 2, 1), (1, 33, 31, 64, 128, 3, 1, 6, 6),
            (8, 64, 64, 64, 128, 3, 1, 1, 1), (4, 32, 32, 256, 256, 3, 1, 1, 1), (2, 16, 16, 512, 512, 3, 2, 1, 1),
            (1, 17, 17, 128, 192, (1, 7), 1, (0, 3), 1)]


@pytest.mark.parametrize("case", WG_CASES)
def test_plane_fed_weight_gradient_on_the_wide_tile_is_bit_identical(tiles, case):
    """dW from dy and x as piece planes: conv_wgw_kernel (64 x 64 per wave, global -> LDS directly, source-side XOR instead of
    row padding under the transposed reads) against conv_wgrad_kernel (dspn_conv_set_wide_tiles(1)) -- same split plan, same
    pixel blocks and piece products per accumulator: the same bits, in the summed gradient and slab by slab; and both against
    the fp32-MFMA weight gradient of the decoded operands."""
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    kh, kw = fn._hw(k); ph, pw = fn._hw(pad)
    g = torch.Generator().manual_seed(H + Cin + Cout + kh + 5)
    Ho, Wo = (H + 2 * ph - dil * (kh - 1) - 1) // stride + 1, (W + 2 * pw - dil * (kw - 1) - 1) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).cuda().abs_()
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    xp, xa = planes_of(x)
    dyp, dya = planes_of(dy)
    wshape = (Cout, kh, kw, Cin)
    kw_ = dict(x_absmax=xa, dy_absmax=dya, x_planes=True, dy_planes=True)
    got = {}
    for mode in (1, 0):
        tiles(mode)
        got[mode] = fn.conv2d_wgrad(xp, dyp, wshape, stride, pad, dil, **kw_)
    assert torch.equal(got[0], got[1])
    splits = fn.conv2d_wgrad_splits(tuple(x.shape), tuple(dy.shape), wshape, stride)
    if splits > 0:
        slabs = {}
        for mode in (1, 0):
            tiles(mode)
            slabs[mode] = torch.full((splits, Cout * kh * kw * Cin), float("nan"), device="cuda")
            fn.conv2d_wgrad_slabs(xp, dyp, wshape, slabs[mode], stride, pad, dil, **kw_)
        assert torch.equal(slabs[0], slabs[1]) and bool(torch.isfinite(slabs[0]).all())
    ref = fn.conv2d_wgrad(x, dy, wshape, stride, pad, dil, math="fp32")
    assert float((got[0] - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
