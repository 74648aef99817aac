"""The `*_bf16` kernels (include/dspn_nn.h: bfloat16 TENSORS in HBM, fp32 arithmetic) against float64 references of the
same operations on the same bf16-representable inputs.

Bars: a result that is STORED as bf16 may differ from the float64 reference by its own rounding (2^-9 relative) plus the
fp32 accumulation error (1e-5 of the output scale); a float32 result (weight gradients, statistics, sums) is held to
1e-5 as in the float build.  Where a statistic describes a stored tensor (BatchNorm statistics in the convolution
epilogue) it must describe the ROUNDED tensor: it is compared with float64 statistics of the device's own output."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from dspnet_amd import functional as fn

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rb(t):
    """round to bf16, keep as float64"""
    return t.float().bfloat16().double()


def nhwc_h(t):  # NCHW cpu double (bf16-representable) -> NHWC cuda bf16, channels padded to 8
    n, c, h, w = t.shape
    out = torch.zeros(n, h, w, fn.padc(c, BF), dtype=BF)
    out[..., :c] = t.permute(0, 2, 3, 1).to(BF)
    return out.cuda()


def nchw(t, c=None):
    t = t.cpu().double().permute(0, 3, 1, 2)
    return t if c is None else t[:, :c]


def wdev32(w):  # [Cout,Cin,R,S] cpu -> float32 master [Cout,R,S,pad8(Cin)] cuda
    co, ci, r, s = w.shape
    out = torch.zeros(co, r, s, fn.padc(ci, BF), dtype=torch.float32)
    out[..., :ci] = w.permute(0, 2, 3, 1).float()
    return out.cuda()


def close_stored(got, exp, what=""):
    """got: device tensor that went through a bf16 store; exp: float64 reference"""
    scale = float(exp.abs().max()) + 1e-30
    bound = exp.abs() * 2.0 ** -8 + 1e-5 * scale
    bad = (got - exp).abs() > bound
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements off, worst {float(((got - exp).abs() - bound).max()):.3e} (scale {scale:.3e})"


def close(got, exp, tol=1e-5):
    scale = float(exp.abs().max()) + 1e-30
    err = float((got - exp).abs().max())
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


CASES = [
    # N, H, W, Cin, Cout, (kh, kw), stride, (ph, pw), dil
    (2, 16, 16, 64, 64, (3, 3), 1, (1, 1), 1),      # backbone 3x3, Cin % 64 == 0: taps stay uniform per k-step
    (2, 17, 19, 32, 48, (3, 3), 2, (1, 1), 1),      # Cin 32: a 64-wide k-step straddles two taps
    (3, 16, 16, 64, 256, (1, 1), 1, (0, 0), 1),
    (2, 16, 16, 128, 256, (1, 1), 2, (0, 0), 1),
    (2, 32, 32, 3, 64, (7, 7), 2, (3, 3), 1),       # conv0: Cin 3 -> 8
    (1, 20, 20, 64, 96, (3, 3), 1, (6, 6), 6),      # dilated
    (2, 8, 8, 256, 20, (3, 3), 1, (1, 1), 1),       # loc head (Cout 20 -> 24)
    (2, 8, 8, 256, 54, (3, 3), 1, (1, 1), 1),       # cls head (Cout 54 -> 56)
    (2, 5, 5, 128, 130, (3, 3), 2, (1, 1), 1),
    (1, 3, 3, 128, 128, (3, 3), 2, (1, 1), 1),
    (4, 64, 64, 64, 128, (3, 3), 1, (1, 1), 1),     # 128 x 128 tiles
    (2, 17, 17, 48, 64, (5, 5), 1, (2, 2), 1),      # inception 5x5 on 48 channels
    (2, 17, 17, 80, 192, (1, 7), 1, (0, 3), 1),     # 1x7
    (2, 17, 17, 80, 192, (7, 1), 1, (3, 0), 1),     # 7x1
    (2, 35, 35, 96, 96, (3, 3), 2, (0, 0), 1),      # 3x3 s2 p0
    (32, 32, 32, 64, 256, (1, 1), 1, (0, 0), 1),    # many tiles, 8-wave variants
]


@pytest.mark.parametrize("case", CASES)
def test_conv_forward_dgrad_wgrad_bf16_tensors(gpu_device, case):
    N, H, W, Cin, Cout, (kh, kw), stride, (ph, pw), dil = case
    g = torch.Generator().manual_seed(sum(map(lambda v: v if isinstance(v, int) else sum(v), case)) + 3)
    x = rb(torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)).requires_grad_()
    w = rb(torch.randn(Cout, Cin, kh, kw, generator=g, dtype=torch.float64) / np.sqrt(Cin * kh * kw)).requires_grad_()
    y_ref = F.conv2d(x, w, None, stride=stride, padding=(ph, pw), dilation=dil)
    dy = rb(torch.randn(y_ref.shape, generator=g, dtype=torch.float64))
    y_ref.backward(dy)
    xd, w32, dyd = nhwc_h(x.detach()), wdev32(w.detach()), nhwc_h(dy)
    wt = fn.weight_transpose(w32, dtype=BF, copy=(wh := torch.empty_like(w32, dtype=BF)))
    assert torch.equal(wh.float(), w32)                           # the copy of a bf16-representable master is exact
    assert wt.dtype == BF and wt.shape == (w32.shape[3], kh, kw, fn.padc(Cout, BF))
    y = fn.conv2d_forward(xd, wh, None, stride=stride, pad=(ph, pw), dil=dil)
    assert y.dtype == BF and y.shape[3] == fn.padc(Cout, BF)
    close_stored(nchw(y, Cout), y_ref.detach(), "forward")
    assert float(y[..., Cout:].float().abs().max() if y.shape[3] > Cout else 0.0) == 0.0
    if stride == 1 or dil == 1:
        dx = fn.conv2d_dgrad(dyd, wt, tuple(xd.shape), stride=stride, pad=(ph, pw), dil=dil)
        close_stored(nchw(dx, Cin), x.grad, "dgrad")
        acc = fn.conv2d_dgrad(dyd, wt, tuple(xd.shape), stride=stride, pad=(ph, pw), dil=dil, out=dx.clone(), accumulate=True)
        close_stored(nchw(acc, Cin), nchw(dx, Cin) + x.grad, "dgrad accumulate")
    dw = fn.conv2d_wgrad(xd, dyd, tuple(w32.shape), stride=stride, pad=(ph, pw), dil=dil)
    assert dw.dtype == torch.float32
    close(dw.cpu().double().permute(0, 3, 1, 2)[:, :Cin], w.grad)
    assert float(dw[..., Cin:].abs().max() if dw.shape[3] > Cin else 0.0) == 0.0


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1, 1), (2, 17, 19, 32, 48, 3, 2, 1), (3, 16, 16, 64, 256, 1, 1, 0),
                                  (1, 9, 9, 40, 40, 3, 1, 1), (4, 64, 64, 64, 128, 3, 1, 1), (32, 32, 32, 64, 256, 1, 1, 0)])
@pytest.mark.parametrize("relu", [True, False])
def test_conv_with_input_affine_bf16_tensors(gpu_device, case, relu):
    """the folded BatchNorm-apply(+ReLU) loader on bf16 tensors: widen, fmaf, (ReLU), zero padding AFTER the affine,
    round to bf16 -- forward and weight gradient"""
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = rb(torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64))
    sc = (torch.rand(Cin, generator=g, dtype=torch.float64) + 0.5).float().double()
    sh = torch.randn(Cin, generator=g, dtype=torch.float64).float().double()
    w = rb(torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)).requires_grad_()
    u = (x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).float().double()       # one rounding, like fmaf
    u = rb(u.clamp_min(0) if relu else u)
    y_ref = F.conv2d(u, w, None, stride=stride, padding=pad)
    dy = rb(torch.randn(y_ref.shape, generator=g, dtype=torch.float64))
    y_ref.backward(dy)
    cp = fn.padc(Cin, BF)
    aff = (torch.cat([sc.float(), torch.zeros(cp - Cin)]).cuda(), torch.cat([sh.float(), torch.zeros(cp - Cin)]).cuda(), relu)
    xd, w32 = nhwc_h(x), wdev32(w.detach())
    wh = w32.to(BF)
    y = fn.conv2d_forward(xd, wh, None, stride=stride, pad=pad, in_affine=aff)
    close_stored(nchw(y, Cout), y_ref.detach(), "forward")
    dw = fn.conv2d_wgrad(xd, nhwc_h(dy), tuple(w32.shape), stride=stride, pad=pad, in_affine=aff)
    close(dw.cpu().double().permute(0, 3, 1, 2)[:, :Cin], w.grad)


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1, 1), (3, 16, 16, 64, 256, 1, 1, 0), (4, 64, 64, 64, 128, 3, 1, 1),
                                  (32, 32, 32, 64, 256, 1, 1, 0), (8, 128, 128, 16, 64, 1, 1, 0)])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_epilogue_statistics_describe_the_stored_bf16_tensor(gpu_device, case, with_res):
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = rb(torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64))
    w = rb(torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g) * 10.0
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, Ho, Wo, Cout, generator=g).to(BF).cuda() if with_res else None
    tiles, tile_rows = fn.conv_stats_layout(N * Ho * Wo, Cout)
    assert tiles > 0
    st = torch.full((tiles, 2, Cout), float("nan"), device="cuda")
    xd, wh = nhwc_h(x), wdev32(w).to(BF)
    y = fn.conv2d_forward(xd, wh, b.cuda(), stride=stride, pad=pad, residual=res, out_stats=st)
    ref = F.conv2d(x, w, b.double(), stride=stride, padding=pad)
    if with_res:
        ref = ref + nchw(res)
    close_stored(nchw(y, Cout), ref, "forward")
    outs = [torch.empty(Cout, device="cuda") for _ in range(4)]
    fn.bn_stats_from_tiles(st, tiles, tile_rows, N * Ho * Wo, Cout, 2e-5, None, torch.zeros(Cout, device="cuda"), *outs)
    yd = y.double().view(-1, y.shape[3])[:, :Cout]
    mean_ref, var_ref = yd.mean(0), yd.var(0, unbiased=False)
    assert float((outs[0].double() - mean_ref).abs().max()) <= 1e-6 * float(mean_ref.abs().max()) + 1e-6
    assert float((outs[1].double() * torch.sqrt(var_ref + 2e-5) - 1).abs().max()) <= 5e-6


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1, 1), (3, 16, 16, 256, 64, 1, 1, 0), (2, 17, 19, 32, 48, 3, 2, 1),
                                  (4, 64, 64, 128, 64, 3, 1, 1), (32, 32, 32, 256, 64, 1, 1, 0)])
def test_dgrad_epilogue_batchnorm_sums_bf16_tensors(gpu_device, case):
    """sums[t] of dspn_conv2d_dgrad_bn_bf16 = per-tile sums of dy' and dy' * xhat over the STORED bf16 dx"""
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 13)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).to(BF).cuda()                  # BatchNorm input
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(BF).cuda()
    w32 = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).to(BF).float().cuda()
    mean = x.float().view(-1, Cin).mean(0)
    rstd = 1.0 / torch.sqrt(x.float().view(-1, Cin).var(0, unbiased=False) + 2e-5)
    scale, shift = rstd.clone(), (-mean * rstd)
    wt = fn.weight_transpose(w32, dtype=BF)
    tiles = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
    sums = torch.full((tiles, 2, Cin), float("nan"), device="cuda")
    d = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, bn_bwd=(x, scale, shift, mean, rstd, True, sums))
    d_plain = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1)
    assert torch.equal(d, d_plain)
    assert torch.isfinite(sums).all()
    dd, xx = d.double().view(-1, Cin), x.double().view(-1, Cin)
    mask = (torch.addcmul(shift.double(), xx, scale.double()).float() > 0).double()
    gsum = (dd * mask).sum(0)
    gxh = (dd * mask * ((xx - mean.double()) * rstd.double())).sum(0)
    got = sums.double().sum(0)
    assert float((got[0] - gsum).abs().max()) <= 2e-5 * float(gsum.abs().max()) + 1e-4
    assert float((got[1] - gxh).abs().max()) <= 2e-5 * float(gxh.abs().max()) + 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# The HBM-bound kernels: each `*_bf16` twin runs the SAME fp32 arithmetic as its `*_f32` original on widened inputs and
# rounds once on store.  On bf16-representable inputs therefore: a stored result == round_to_bf16(float result) bit
# for bit, and a float result (statistics, column sums, probabilities) == the float kernel's, bit for bit.
def _pair(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    h = (torch.randn(*shape, generator=g) * scale).to(BF).cuda()
    return h, h.float()


def _same_stored(h_out, f_out, what):
    assert h_out.dtype == BF and f_out.dtype == torch.float32
    assert torch.equal(h_out, f_out.to(BF)), f"{what}: bf16 kernel != round(float kernel), max diff {float((h_out.float() - f_out).abs().max()):.3e}"


def test_batchnorm_kernels_bf16_twins(gpu_device):
    for shape, relu in (((4, 16, 16, 64), True), ((2, 9, 7, 256), False), ((3, 5, 5, 24), True)):
        xh, xf = _pair(shape, 1)
        dyh, dyf = _pair(shape, 2)
        C = shape[-1]
        gamma = (torch.rand(C) + 0.5).cuda(); beta = torch.randn(C).cuda()
        sh_, sf_ = fn.bn_stats(xh, 2e-5, gamma, beta), fn.bn_stats(xf, 2e-5, gamma, beta)
        for a, b in zip(sh_, sf_):
            assert torch.equal(a, b)                                    # statistics: float, same sums in the same order
        mean, rstd, scale, shift = sf_
        _same_stored(fn.bn_apply(xh, scale, shift, relu=relu), fn.bn_apply(xf, scale, shift, relu=relu), "bn_apply")
        for acc in (False, True):
            baseh, basef = _pair(shape, 3)
            dxh, dgh, dbh = fn.bn_backward(xh, scale, shift, dyh, mean, rstd, gamma, relu=relu, dx=baseh.clone(), accumulate=acc)
            dxf, dgf, dbf = fn.bn_backward(xf, scale, shift, dyf, mean, rstd, gamma, relu=relu, dx=basef.clone(), accumulate=acc)
            _same_stored(dxh, dxf, "bn_backward dx")
            assert torch.equal(dgh, dgf) and torch.equal(dbh, dbf)


def test_elementwise_pool_layout_kernels_bf16_twins(gpu_device):
    ah, af = _pair((2, 16, 16, 32), 4)
    bh, bf_ = _pair((2, 16, 16, 32), 5)
    _same_stored(fn.add(ah, bh), fn.add(af, bf_), "add")
    yh, yf = ah.clamp(min=0), af.clamp(min=0)
    _same_stored(fn.relu_backward(yh, bh), fn.relu_backward(yf, bf_), "relu_backward")
    dxh, sh_ = fn.relu_backward_colsum(yh.view(-1, 32), bh.view(-1, 32).clone(), 30)
    dxf, sf_ = fn.relu_backward_colsum(yf.view(-1, 32), bf_.view(-1, 32).clone(), 30)
    _same_stored(dxh, dxf, "relu_backward_colsum dx")
    assert torch.equal(sh_, sf_)
    assert torch.equal(fn.colsum(ah.view(-1, 32), 30), fn.colsum(af.view(-1, 32), 30))
    # pooling
    am_h = torch.zeros(2, 8, 8, 32, dtype=torch.uint8, device="cuda"); am_f = torch.zeros_like(am_h)
    ph, pf = fn.maxpool_forward(ah, 3, 2, 1, argmax=am_h), fn.maxpool_forward(af, 3, 2, 1, argmax=am_f)
    _same_stored(ph, pf, "maxpool")
    assert torch.equal(am_h, am_f)
    gh, gf = _pair((2, 8, 8, 32), 6)
    _same_stored(fn.maxpool_backward_argmax(am_h, gh, ah.shape, 3, 2, 1), fn.maxpool_backward_argmax(am_f, gf, af.shape, 3, 2, 1), "maxpool_bwd_idx")
    _same_stored(fn.maxpool_backward(ah, ph, gh, 3, 2, 1), fn.maxpool_backward(af, pf, gf, 3, 2, 1), "maxpool_bwd")
    _same_stored(fn.avgpool_forward(ah, 2), fn.avgpool_forward(af, 2), "avgpool")
    _same_stored(fn.avgpool_backward(gh, ah.shape, 2), fn.avgpool_backward(gf, af.shape, 2), "avgpool_bwd")
    _same_stored(fn.avgpool2d_forward(ah, 3, 1, 1), fn.avgpool2d_forward(af, 3, 1, 1), "avgpool2d")
    _same_stored(fn.avgpool2d_backward(ah, ah.shape, 3, 1, 1), fn.avgpool2d_backward(af, af.shape, 3, 1, 1), "avgpool2d_bwd")
    # tap sum / spread
    zh, zf = _pair((2, 8, 8, 48), 7)
    bias = torch.randn(5).cuda()
    oh, of = torch.zeros(2, 8, 8, 8, dtype=BF, device="cuda"), torch.zeros(2, 8, 8, 8, device="cuda")
    _same_stored(fn.tap_sum(zh, bias, 5, 3, 3, (1, 1), oh), fn.tap_sum(zf, bias, 5, 3, 3, (1, 1), of), "tap_sum")
    sh2, sf2 = torch.zeros(2, 8, 8, 48, dtype=BF, device="cuda"), torch.zeros(2, 8, 8, 48, device="cuda")
    _same_stored(fn.tap_spread(oh, 5, 3, 3, (1, 1), sh2), fn.tap_spread(of, 5, 3, 3, (1, 1), sf2), "tap_spread")
    # layout: float NCHW image -> NHWC of either storage type; block copies within and across storage types
    img = torch.randn(2, 3, 9, 7).to(BF).float().cuda()
    nh = fn.nchw_to_nhwc(img, out=torch.empty(2, 9, 7, 8, dtype=BF, device="cuda"))
    nf = fn.nchw_to_nhwc(img, out=torch.empty(2, 9, 7, 8, device="cuda"))
    _same_stored(nh, nf, "nchw_to_nhwc")
    src_h, src_f = _pair((2, 6, 6, 16), 8)
    for acc in (False, True):
        dh, df = _pair((2, 6 * 6 * 10 + 7,), 9)
        dh, df = dh.view(2, -1), df.view(2, -1)
        f32_out = df.clone()
        fn.copy_block(src_f, f32_out, 2, 36, 10, 36 * 16, 16, 3, f32_out.shape[1], 10, 5, accumulate=acc)
        h_out = dh.clone()
        fn.copy_block(src_h, h_out, 2, 36, 10, 36 * 16, 16, 3, h_out.shape[1], 10, 5, accumulate=acc)       # bf16 -> bf16
        _same_stored(h_out, f32_out, "copy_block bf16->bf16")
        mixed = df.clone()
        fn.copy_block(src_h, mixed, 2, 36, 10, 36 * 16, 16, 3, mixed.shape[1], 10, 5, accumulate=acc)       # bf16 -> float
        assert torch.equal(mixed, f32_out)
        back = dh.clone()
        fn.copy_block(src_f, back, 2, 36, 10, 36 * 16, 16, 3, back.shape[1], 10, 5, accumulate=acc)         # float -> bf16
        _same_stored(back, f32_out, "copy_block float->bf16")


def test_softmax_output_and_affine_sampler_bf16_twins(gpu_device):
    lh, lf = _pair((500, 24), 10, 3.0)
    label = torch.randint(0, 19, (500,)).float().cuda(); label[::7] = 255.0
    ph, gh = fn.softmax_output(lh, label, 19, 255.0, 0.25)
    pf, gf = fn.softmax_output(lf, label, 19, 255.0, 0.25)
    assert ph.dtype == torch.float32 and torch.equal(ph, pf)
    _same_stored(gh, gf, "softmax gradient")
    theta = torch.tensor([0.97, 0.02, 0.01, -0.03, 1.04, 0.02], device="cuda")
    srcs_h, srcs_f = zip(*[_pair((2, h, w, 8), 20 + h) for h, w in ((4, 4), (8, 8), (16, 12))])
    outh, outf = torch.empty(2, 16, 12, 8, dtype=BF, device="cuda"), torch.empty(2, 16, 12, 8, device="cuda")
    fn.affine_sampler_forward(fn.SamplerSources([(t, 0) for t in srcs_h]), theta, outh)
    fn.affine_sampler_forward(fn.SamplerSources([(t, 0) for t in srcs_f]), theta, outf)
    _same_stored(outh, outf, "sampler forward")
    dyh, dyf = _pair((2, 16, 12, 8), 30)
    for th_, tf_ in zip(srcs_h, srcs_f):
        _same_stored(fn.affine_sampler_backward_data(dyh, theta, th_.shape, 0), fn.affine_sampler_backward_data(dyf, theta, tf_.shape, 0),
                     "sampler backward data")
    dth, dtf = torch.zeros(6, device="cuda"), torch.zeros(6, device="cuda")
    fn.affine_sampler_backward_theta(fn.SamplerSources([(t, 0) for t in srcs_h]), theta, dyh, dth)
    fn.affine_sampler_backward_theta(fn.SamplerSources([(t, 0) for t in srcs_f]), theta, dyf, dtf)
    # (float sums of the same terms; the two builds may contract the per-lane multiply-adds differently)
    assert float((dth - dtf).abs().max()) <= 1e-6 * float(dtf.abs().max())
