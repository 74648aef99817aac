"""GPU numerics of the conv / BN / pooling / sampler / loss / SGD kernels (through the C ABI)
against plain PyTorch CPU references of the same ops in float64.

These ops are MXNet built-ins in the reference (not vendored: "parity unpinned"); the semantics
checked here are the documented MXNet ones restated in oracle/dspnet_torch.py.  Tolerance: the fp32
MFMA path is an exact fmaf chain, so |err| <= 2e-6 * sum|a*b| is expected; tests use 1e-4 relative to
the output scale (BASELINE.json: fp32 losses within 1e-4)."""
import numpy as np
import math
import pytest
import torch
import torch.nn.functional as F

from dspnet_amd import functional as fn

pytestmark = pytest.mark.gpu


def nhwc(t):  # NCHW cpu double -> NHWC cuda float (channel padded to 4)
    n, c, h, w = t.shape
    cp = fn.pad4(c)
    out = torch.zeros(n, h, w, cp, dtype=torch.float32)
    out[..., :c] = t.permute(0, 2, 3, 1).float()
    return out.cuda()


def nchw(t, c=None):  # NHWC cuda -> NCHW cpu double
    t = t.cpu().double().permute(0, 3, 1, 2)
    return t if c is None else t[:, :c]


def wdev(w):  # [Cout,Cin,R,S] cpu -> [Cout,R,S,pad4(Cin)] cuda
    co, ci, r, s = w.shape
    out = torch.zeros(co, r, s, fn.pad4(ci), dtype=torch.float32)
    out[..., :ci] = w.permute(0, 2, 3, 1).float()
    return out.cuda()


def close(got, exp, tol=1e-4):
    scale = float(exp.abs().max()) + 1e-30
    err = float((got - exp).abs().max())
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.fixture(params=["bf16x3", "fp32", "f16x2"])
def conv_math(request):
    """the two math modes of float-tensor convolutions that promise fp32 results -- DSPN_MATH_F32_BF16X3 (three-piece bf16
    split, six exact products per multiply, fp32 accumulate: the default) and DSPN_MATH_FP32 (fp32 MFMA) -- held to the same
    tolerances"""
    fn.set_conv_math(request.param)
    yield request.param
    fn.set_conv_math(fn.DEFAULT_CONV_MATH)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 16, 16, 64, 64, 3, 1, 1, 1),      # backbone 3x3
    (2, 17, 19, 32, 48, 3, 2, 1, 1),      # stage-entry 3x3 s2, odd sizes
    (3, 16, 16, 64, 256, 1, 1, 0, 1),     # 1x1 expand
    (2, 16, 16, 128, 256, 1, 2, 0, 1),    # shortcut 1x1 s2
    (2, 32, 32, 3, 64, 7, 2, 3, 1),       # conv0 (Cin 3 -> padded 4)
    (1, 20, 20, 64, 96, 3, 1, 6, 6),      # vgg fc6 dilated
    (2, 8, 8, 256, 20, 3, 1, 1, 1),       # loc head (Cout 20)
    (2, 8, 8, 256, 54, 3, 1, 1, 1),       # cls head (Cout 54, not a multiple of 4)
    (1, 16, 16, 200, 19, 3, 1, 1, 1),     # score3_conv-like (Cout 19)
    (2, 5, 5, 128, 130, 3, 2, 1, 1),      # extras, tiny maps
    (1, 3, 3, 128, 128, 3, 2, 1, 1),
    (4, 64, 64, 64, 128, 3, 1, 1, 1),     # 512 tiles of 64x64
    # 3x3 / stride 1 layers large enough for every tile shape of conv_nt_kernel without split-K (the split math reads the
    # weights of all of these as piece planes)
    (8, 64, 64, 64, 128, 3, 1, 1, 1),     # forward: 128x128 on 8 waves; data gradient: 64x64
    (8, 64, 64, 128, 128, 3, 1, 1, 1),    # both on 8 waves
    (4, 64, 64, 128, 192, 3, 1, 1, 1),    # forward: 128x64 tiles (Cout just past 128)
    (32, 16, 16, 64, 128, 3, 1, 1, 1),    # stage-4 sized maps
    (3, 20, 32, 64, 96, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_dgrad_wgrad(gpu_device, conv_math, case):
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)).requires_grad_()
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    y_ref = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)

    xd, wd_, bd = nhwc(x.detach()), wdev(w.detach()), b.float().cuda()
    y = fn.conv2d_forward(xd, wd_, bd, stride=stride, pad=pad, dil=dil)
    close(nchw(y, Cout), y_ref.detach())
    if y.shape[3] != Cout:
        assert float(y[..., Cout:].abs().max()) == 0.0      # pad channels untouched (zero)
    # relu epilogue
    yr = fn.conv2d_forward(xd, wd_, bd, stride=stride, pad=pad, dil=dil, relu=True)
    close(nchw(yr, Cout), y_ref.detach().clamp(min=0))

    dyd = nhwc(dy)
    if stride == 1 or dil == 1:
        wt = fn.weight_transpose(wd_)
        dx = fn.conv2d_dgrad(dyd, wt, tuple(xd.shape), stride=stride, pad=pad, dil=dil)
        close(nchw(dx, Cin), x.grad)
        dx2 = fn.conv2d_dgrad(dyd, wt, tuple(xd.shape), stride=stride, pad=pad, dil=dil, out=dx.clone(),
                              accumulate=True)
        close(nchw(dx2, Cin), 2 * x.grad)
    dw = fn.conv2d_wgrad(xd, dyd, tuple(wd_.shape), stride=stride, pad=pad, dil=dil)
    close(dw.cpu().double().permute(0, 3, 1, 2)[:, :Cin], w.grad)
    close(fn.colsum(dyd, Cout).cpu().double(), dy.sum(dim=(0, 2, 3)))


@pytest.mark.parametrize("shape", [(64, 3, 3, 64), (19, 1, 1, 128), (40, 3, 3, 32), (171, 1, 1, 2048), (96, 1, 7, 160)])
def test_weight_planes_layout_and_pieces(gpu_device, shape):
    """dspn_conv2d_weight_planes_f32: planes[row][tap][cols / 32][piece][32] with p0 = bf16(x), p1 = bf16(x - p0),
    p2 = bf16(x - p0 - p1), of the weight itself and of its zero-padded transpose; the batch form writes the same bits"""
    Cout, R, S, Cin = shape
    g = torch.Generator().manual_seed(sum(shape))
    w = (torch.randn(shape, generator=g) * torch.exp(4 * torch.randn(shape, generator=g))).cuda()

    def pieces(m):      # m (rows, taps, cols) float32 -> (rows, taps, cols / 32, 3, 32) bfloat16
        p0 = m.bfloat16(); r1 = m - p0.float()
        p1 = r1.bfloat16(); p2 = (r1 - p1.float()).bfloat16()
        st = torch.stack([p0, p1, p2], dim=0)                      # (3, rows, taps, cols)
        rows, taps, cols = m.shape
        return st.view(3, rows, taps, cols // 32, 32).permute(1, 2, 3, 0, 4).contiguous()

    fwd = fn.weight_planes(w, math="bf16x3")
    assert torch.equal(fwd.view(torch.int16), pieces(w.view(Cout, R * S, Cin)).view(torch.int16))
    cols = (Cout + 31) // 32 * 32
    wt = torch.zeros(Cin, R * S, cols, device="cuda")
    wt[:, :, :Cout] = w.view(Cout, R * S, Cin).permute(2, 1, 0)
    bwd = fn.weight_planes(w, transposed=True, cols=cols, math="bf16x3")
    assert torch.equal(bwd.view(torch.int16), pieces(wt).view(torch.int16))
    # p0 + p1 + p2 reproduces the float to half an fp32 ulp
    back = fwd.float().sum(dim=3).view(Cout, R * S, Cin)
    assert float(((back - w.view(Cout, R * S, Cin)).abs() / w.view(Cout, R * S, Cin).abs().clamp(min=1e-30)).max()) <= 2.0 ** -23
    a, b = torch.zeros_like(fwd), torch.zeros_like(bwd)
    fn.weight_planes_batch(*fn.weight_planes_table([(w, a, b), (w, None, torch.zeros_like(b)), (w, torch.zeros_like(a), None)],
                                                   w.device))
    assert torch.equal(a.view(torch.int16), fwd.view(torch.int16)) and torch.equal(b.view(torch.int16), bwd.view(torch.int16))

    # DSPN_MATH_F32_F16X2: two float16 pieces of w * 2^e, 2^e the power of two that puts max |w| into [2^14, 2^15)
    am = fn.absmax(w)
    e = 15 - math.frexp(float(w.abs().max()))[1]
    assert float(am.max()) == float(w.abs().max())

    def pieces2(m):
        u = m * 2.0 ** e
        h0 = u.half(); h1 = (u - h0.float()).half()
        rows, taps, cols_ = m.shape
        return torch.stack([h0, h1], dim=0).view(2, rows, taps, cols_ // 32, 32).permute(1, 2, 3, 0, 4).contiguous()

    fwd2 = fn.weight_planes(w, math="f16x2", w_absmax=am)
    assert fwd2.shape == (Cout, R * S, Cin // 32, 2, 32)
    assert torch.equal(fwd2.view(torch.int16), pieces2(w.view(Cout, R * S, Cin)).view(torch.int16))
    bwd2 = fn.weight_planes(w, transposed=True, cols=cols, math="f16x2", w_absmax=am)
    assert torch.equal(bwd2.view(torch.int16), pieces2(wt).view(torch.int16))
    a2, b2 = torch.zeros_like(fwd2), torch.zeros_like(bwd2)
    fn.weight_planes_batch(*fn.weight_planes_table([(w, a2, b2, am), (w, a, None)], w.device))
    assert torch.equal(a2.view(torch.int16), fwd2.view(torch.int16)) and torch.equal(b2.view(torch.int16), bwd2.view(torch.int16))
    assert torch.equal(a.view(torch.int16), fwd.view(torch.int16))          # a three-piece row beside a two-piece one
    with pytest.raises(AssertionError):
        fn.weight_planes(w, math="f16x2")                                   # no magnitude block


def test_split_math_needs_planes_at_the_c_abi(gpu_device):
    """DSPN_MATH_F32_BF16X3 with Cin % 32 == 0 and no piece planes is an argument error, not a silent slow path"""
    from dspnet_amd._lib import DspnError
    x = torch.zeros(1, 8, 16, 64, device="cuda"); w = torch.zeros(64, 3, 3, 64, device="cuda"); y = torch.zeros(1, 8, 16, 64, device="cuda")
    L = fn.L()
    rc = L.dspn_conv2d_forward_bn_f32(fn.ptr(x), 0, 0, 0, fn.ptr(w), 0, 0, 0, fn.ptr(y), 1, 8, 16, 64, 64, 3, 3, 1, 1, 1, 1, 8, 16,
                                      0, 64, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, fn.stream())
    assert rc != 0 and b"piece planes" in L.dspn_last_error()
    with pytest.raises(DspnError):
        fn.check(rc, "conv2d_forward")


def test_conv_large_k_and_split_k_determinism(gpu_device, conv_math):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 832, 16, 16, generator=g, dtype=torch.float64)
    w = torch.randn(19, 832, 3, 3, generator=g, dtype=torch.float64) / 80
    dy = torch.randn(2, 19, 16, 16, generator=g, dtype=torch.float64)
    xd, wd_, dyd = nhwc(x), wdev(w), nhwc(dy)
    y = fn.conv2d_forward(xd, wd_, pad=1)
    close(nchw(y, 19), F.conv2d(x, w, padding=1))
    a = fn.conv2d_wgrad(xd, dyd, tuple(wd_.shape), pad=1)
    b = fn.conv2d_wgrad(xd, dyd, tuple(wd_.shape), pad=1)
    assert torch.equal(a, b)
    ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1)
    close(a.cpu().double().permute(0, 3, 1, 2), ref)


def test_deconv_4x4_s2_forward_backward(gpu_device, conv_math):
    """mx.sym.Deconvolution(kernel 4, stride 2, pad 1, no bias) == dgrad of a 4x4/2 conv"""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 19, 12, 10, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(19, 19, 4, 4, generator=g, dtype=torch.float64, requires_grad=True)  # [Cin, Cout, kh, kw]
    y_ref = F.conv_transpose2d(x, w, stride=2, padding=1)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    # device weight is the [K=Cin_deconv][R][S][C=Cout_deconv] tensor of the conv whose dgrad this is
    wk = torch.zeros(20, 4, 4, 20)
    wk[:19, :, :, :19] = w.detach().permute(0, 2, 3, 1).float()
    wk = wk.cuda()
    xd = nhwc(x.detach())                     # (2,12,10,20)
    wt = fn.weight_transpose(wk)              # [C][R][S][K]
    y = fn.conv2d_dgrad(xd, wt, (2, 24, 20, 20), stride=2, pad=1)
    close(nchw(y, 19), y_ref.detach())
    dyd = nhwc(dy)
    dx = fn.conv2d_forward(dyd, wk, stride=2, pad=1)          # backward-data of deconv = conv forward
    close(nchw(dx, 19), x.grad)
    dw = fn.conv2d_wgrad(dyd, xd, tuple(wk.shape), stride=2, pad=1)   # roles swapped
    close(dw[:19, :, :, :19].cpu().double().permute(0, 3, 1, 2), w.grad)


@pytest.mark.parametrize("shape,relu,fix_gamma", [((4, 16, 16, 64), True, False), ((2, 9, 7, 256), True, False),
                                                   ((3, 8, 8, 20), False, True), ((2, 4, 4, 2048), False, True),
                                                   ((8, 64, 64, 4), False, True)])
def test_batchnorm_forward_backward(gpu_device, shape, relu, fix_gamma):
    g = torch.Generator().manual_seed(3)
    C = shape[3]
    x = (torch.randn(shape, generator=g, dtype=torch.float64) * 3 + 5).requires_grad_()
    gamma = torch.ones(C, dtype=torch.float64) if fix_gamma else (torch.rand(C, generator=g, dtype=torch.float64) + 0.5)
    gamma.requires_grad_(not fix_gamma)
    beta = torch.randn(C, generator=g, dtype=torch.float64).requires_grad_()
    eps = 2e-5
    xf = x.reshape(-1, C)
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    y_ref = (x - mean) / torch.sqrt(var + eps) * gamma + beta
    if relu:
        y_ref = y_ref.clamp(min=0)
    dy = torch.randn(shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)

    xd = x.detach().float().cuda()
    gd = None if fix_gamma else gamma.detach().float().cuda()
    bd = beta.detach().float().cuda()
    m, r, sc, sh = fn.bn_stats(xd, eps, gd, bd)
    close(m.cpu().double(), mean.detach(), 1e-6)
    close(r.cpu().double(), 1 / torch.sqrt(var.detach() + eps), 1e-5)
    y = fn.bn_apply(xd, sc, sh, relu=relu)
    close(y.cpu().double(), y_ref.detach(), 1e-5)
    dx, dgam, dbet = fn.bn_backward(xd, sc, sh, dy.float().cuda(), m, r, gd, relu=relu)
    close(dx.cpu().double(), x.grad, 1e-4)
    close(dbet.cpu().double(), beta.grad, 1e-5)
    if not fix_gamma:
        close(dgam.cpu().double(), gamma.grad, 1e-5)
    dx2, _, _ = fn.bn_backward(xd, sc, sh, dy.float().cuda(), m, r, gd, relu=relu, dx=dx.clone(), accumulate=True)
    close(dx2.cpu().double(), 2 * x.grad, 1e-4)


def test_maxpool_3x3_s2_p1(gpu_device):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 8, 13, 14, generator=g, dtype=torch.float64).clamp(min=0).requires_grad_()  # many ties at 0
    y_ref = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    xd = nhwc(x.detach())
    y = fn.maxpool_forward(xd, 3, 2, 1)
    assert torch.equal(nchw(y), y_ref.detach().float().double())
    dx = fn.maxpool_backward(xd, y, nhwc(dy), 3, 2, 1)
    close(nchw(dx), x.grad, 1e-6)
    # the argmax-record path (what the graph uses) must give the same bits
    am = torch.zeros(y.shape, dtype=torch.uint8, device="cuda")
    y2 = fn.maxpool_forward(xd, 3, 2, 1, argmax=am)
    assert torch.equal(y2, y)
    assert torch.equal(fn.maxpool_backward_argmax(am, nhwc(dy), xd.shape, 3, 2, 1), dx)


def test_maxpool_with_folded_batchnorm_relu(gpu_device):
    """Round 4: dspn_maxpool_forward_bn_f32 -- max pooling of relu(x * scale + shift) without the normalised tensor -- gives
    the values, the argmax record and the magnitude of bn_apply followed by the plain pooling, bit for bit (negative scales
    included: the affine is applied per element before the comparison, not to the maximum)"""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(3, 37, 41, 24, generator=g).cuda()
    sc = (torch.randn(24, generator=g)).cuda(); sh = (torch.randn(24, generator=g) * 0.5).cuda()
    for relu in (True, False):
        y = fn.bn_apply(x, sc, sh, relu=relu)
        ref = fn.maxpool_forward(y, 3, 2, 1)
        idx_ref = torch.zeros(ref.shape, dtype=torch.uint8, device="cuda"); fn.maxpool_forward(y, 3, 2, 1, argmax=idx_ref)
        idx = torch.zeros_like(idx_ref); am = torch.zeros(64, device="cuda")
        got = fn.maxpool_forward(x, 3, 2, 1, argmax=idx, in_affine=(sc, sh, relu), out_absmax=am)
        assert torch.equal(got, ref) and torch.equal(idx, idx_ref)
        assert float(am.max()) == float(ref.abs().max())
        assert torch.equal(fn.maxpool_forward(x, 3, 2, 1, in_affine=(sc, sh, relu)), ref)


@pytest.mark.parametrize("k", [1, 2, 4])
def test_avgpool(gpu_device, k):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 8, 16, 16, generator=g, dtype=torch.float64, requires_grad=True)
    y_ref = F.avg_pool2d(x, k, k)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    y = fn.avgpool_forward(nhwc(x.detach()), k)
    close(nchw(y), y_ref.detach(), 1e-6)
    dx = fn.avgpool_backward(nhwc(dy), (2, 16, 16, 8), k)
    close(nchw(dx), x.grad, 1e-6)


@pytest.mark.parametrize("hin,win", [(4, 4), (8, 8), (16, 16), (64, 64), (5, 9)])
def test_bilinear_sampler_identity_grid(gpu_device, hin, win):
    """BilinearSampler(GridGenerator(identity affine)) == align_corners bilinear resize"""
    Ho, Wo = 64, 64
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 8, hin, win, generator=g, dtype=torch.float64, requires_grad=True)
    y_ref = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=True)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    out = torch.zeros(2, Ho, Wo, 24, device="cuda")
    fn.bilinear_forward(nhwc(x.detach()), out, 8)
    close(nchw(out)[:, 8:16], y_ref.detach(), 1e-5)
    assert float(out[..., :8].abs().max()) == 0 and float(out[..., 16:].abs().max()) == 0
    dyc = torch.zeros(2, Ho, Wo, 24, device="cuda")
    dyc[..., 8:16] = nhwc(dy)
    dx = fn.bilinear_backward(dyc, (2, hin, win, 8), 8)                       # separable two-pass kernels
    close(nchw(dx), x.grad, 1e-5)
    dx1 = fn.bilinear_backward(dyc, (2, hin, win, 8), 8, separable=False)     # one-pass gather kernel
    close(nchw(dx1), x.grad, 1e-5)


THETAS = [(1, 0, 0, 0, 1, 0),                                    # multi_init.py:72
          (0.98, 0.03, -0.02, -0.04, 1.05, 0.01),                 # a few SGD steps away from it
          (0.71, 0.29, 0.23, -0.26, 0.83, -0.11),                  # rotation + shear + shift: parts of the grid leave the image
                                                                  # (generic values: round ones put samples exactly on source pixels)
          (1.31, 0.0, 0.0, 0.0, 1.29, 0.0)]                         # zoom out: a border of zero padding


@pytest.mark.parametrize("theta", THETAS)
@pytest.mark.parametrize("shapes", [[(4, 4), (8, 8), (16, 16)], [(16, 16)], [(5, 9), (16, 12)]])
def test_affine_sampler_matches_torch_grid_sample(gpu_device, theta, shapes):
    """GridGenerator(affine_matrix) + BilinearSampler (multitask_symbol_builder.py:574-581) with a learnable
    affine_matrix: forward, data gradients and d/d affine_matrix against affine_grid + grid_sample autograd in
    float64 (align_corners=True, zeros padding); concat mode (disjoint slices) and sum mode (one slice)."""
    Ho, Wo, B, C = 16, 12, 2, 8
    g = torch.Generator().manual_seed(11)
    th = torch.tensor(theta, dtype=torch.float64).requires_grad_()
    xs = [torch.randn(B, C, h, w, generator=g, dtype=torch.float64, requires_grad=True) for h, w in shapes]
    grid = F.affine_grid(th.view(1, 2, 3).expand(B, 2, 3), (B, 1, Ho, Wo), align_corners=True)
    samp = [F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True) for x in xs]
    th_dev = torch.tensor(theta, dtype=torch.float32, device="cuda")
    for mode in ("concat", "sum"):
        for t in xs + [th]:
            t.grad = None
        y_ref = torch.cat(samp, dim=1) if mode == "concat" else sum(samp)
        dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
        y_ref.backward(dy, retain_graph=True)
        offs = [C * i if mode == "concat" else 0 for i in range(len(xs))]
        xd = [nhwc(x.detach()) for x in xs]
        src = fn.SamplerSources(list(zip(xd, offs)))
        out = torch.full((B, Ho, Wo, y_ref.shape[1] + 4), 7.0, device="cuda")        # 4 uncovered channels
        fn.affine_sampler_forward(src, th_dev, out)
        close(nchw(out)[:, :y_ref.shape[1]], y_ref.detach(), 1e-5)
        assert float(out[..., y_ref.shape[1]:].abs().max()) == 0
        dyc = torch.zeros_like(out)
        dyc[..., :y_ref.shape[1]] = nhwc(dy)[..., :y_ref.shape[1]]
        for x, xdev, off in zip(xs, xd, offs):
            dx = fn.affine_sampler_backward_data(dyc, th_dev, xdev.shape, off)
            close(nchw(dx), x.grad, 1e-5)
            acc = fn.affine_sampler_backward_data(dyc, th_dev, xdev.shape, off, dx=dx.clone(), accumulate=True)
            close(nchw(acc), 2 * x.grad, 1e-5)
        dth = torch.zeros(6, device="cuda")
        fn.affine_sampler_backward_theta(src, th_dev, dyc, dth)
        # at the identity grid a source of the target's own height or width is sampled exactly ON its pixels, where the
        # interpolation has a kink (left / right derivative differ): d/d theta is compared off the kinks only
        kink = tuple(theta) == (1, 0, 0, 0, 1, 0) and any(h == Ho or w == Wo for h, w in shapes)
        if not kink:
            close(dth.cpu().double(), th.grad, 1e-4)
        again = torch.zeros(6, device="cuda")
        fn.affine_sampler_backward_theta(src, th_dev, dyc, again)
        assert torch.equal(dth, again)                                   # fixed-order reductions
        # round 4: the same gradient as a by-product of the data-gradient passes (one call per source, rows of float64
        # partial sums, two-level fixed-order reduce) -- with the forward values read from the buffer dx overwrites
        rows = [fn.affine_sampler_theta_rows(xdev.shape, Ho) for xdev in xd]
        part = torch.full((sum(rows), 6), float("nan"), dtype=torch.float64, device="cuda")
        r0 = 0
        for x, xdev, off, r in zip(xs, xd, offs, rows):
            buf = xdev.clone()
            dx = fn.affine_sampler_backward_data_theta(dyc, th_dev, buf, off, part[r0:r0 + r], dx=buf)      # in place
            assert torch.equal(dx, fn.affine_sampler_backward_data(dyc, th_dev, xdev.shape, off))
            r0 += r
        fused = torch.zeros(6, device="cuda")
        fn.affine_sampler_theta_reduce(part, fused)
        if not kink:
            close(fused.cpu().double(), th.grad, 1e-4)
        scale = float(dth.abs().max()) + 1e-6
        assert float((fused - dth).abs().max()) <= 1e-4 * scale, (fused, dth)     # same sum, other order (kinks included)
        again = torch.zeros(6, device="cuda")
        fn.affine_sampler_theta_reduce(part, again)
        assert torch.equal(fused, again)


@pytest.mark.parametrize("hw", [(2, 2), (4, 4), (3, 5)])
def test_affine_sampler_small_source_maps_split_over_workgroups(gpu_device, hw):
    """Round 4: a 2 x 2 or 4 x 4 source sampled to 64 x 64 -- every source pixel gathers from thousands of target pixels -- runs
    its data-gradient / theta pass as 8 workgroups per pixel and one fixed-order reduce; same dx (to rounding: other
    summation order), same theta gradient, in place and accumulating, and bitwise reproducible"""
    g = torch.Generator().manual_seed(21)
    B, C, Ho, Wo = 3, 44, 64, 64
    th = torch.tensor([0.97, 0.04, -0.03, -0.05, 1.04, 0.02], device="cuda")
    x = torch.randn(B, hw[0], hw[1], C, generator=g).cuda()
    dy = torch.randn(B, Ho, Wo, C + 4, generator=g).cuda()
    assert fn.affine_sampler_theta_rows(x.shape, Ho) == 8 * B * hw[0] * hw[1]          # the split form is what runs
    ref = fn.affine_sampler_backward_data(dy, th, x.shape, 4)
    dth_ref = torch.zeros(6, device="cuda")
    fn.affine_sampler_backward_theta(fn.SamplerSources([(x, 4)]), th, dy, dth_ref)
    part = torch.full((fn.affine_sampler_theta_rows(x.shape, Ho), 6), float("nan"), dtype=torch.float64, device="cuda")
    buf = x.clone(); am = torch.zeros(64, device="cuda")
    dx = fn.affine_sampler_backward_data_theta(dy, th, buf, 4, part, dx=buf, dx_absmax=am)      # in place
    assert float((dx - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float(am.max()) == float(dx.abs().max())
    dth = torch.zeros(6, device="cuda")
    fn.affine_sampler_theta_reduce(part, dth)
    assert float((dth - dth_ref).abs().max()) <= 1e-4 * float(dth_ref.abs().max())
    acc = fn.affine_sampler_backward_data_theta(dy, th, x, 4, part, dx=ref.clone(), accumulate=True)
    assert float((acc - 2 * ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    again = fn.affine_sampler_backward_data_theta(dy, th, x, 4, part)
    assert torch.equal(again, dx)


@pytest.mark.parametrize("theta", [(1, 0, 0, 0, 1, 0), (0.97, 0.04, -0.03, -0.05, 1.04, 0.02), (0.7, -0.5, 0.1, 0.45, 0.8, -0.2),
                                   (0.03, 0.01, 0.2, -0.02, 0.04, -0.1), (3.0, 0.0, 0.0, 0.0, 2.5, 0.0)])
@pytest.mark.parametrize("hw", [(64, 64), (32, 32), (16, 16), (33, 20)])
def test_affine_sampler_batched_data_gradient_gives_the_same_bits(gpu_device, theta, hw):
    """Round 6: the data gradient + theta rows with the geometry of a source position taken ONCE for the whole batch
    (sampler_bwd_data_batched_kernel: a workgroup per position lists the matching target pixels in the order the per-pixel
    kernel visits them, every wave walks its share of the batch) against the per-pixel kernel it replaces
    (dspn_affine_sampler_set_batched(0)): dx overwritten / in place / accumulating, the magnitude block and the float64 theta
    rows -- bit for bit on the identity grid, within 2e-5 of the largest entry otherwise -- near-identity grids (a handful of matches per position), a rotation, a strongly MINIFYING theta
    (thousands of matches per position: the list overflows and the position is walked as before) and a magnifying one (most
    positions get no match), sources of the target's size and smaller (the four-slice order of the 16 x 16 level), a
    workgroup count that does not divide the batch."""
    from dspnet_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(hw[0] * 7 + hw[1])
    B, C, Ho, Wo = 5, 172, 64, 64
    th = torch.tensor(theta, dtype=torch.float32, device="cuda")
    x = torch.randn(B, hw[0], hw[1], C, generator=g).cuda()
    dy = torch.randn(B, Ho, Wo, C + 4, generator=g).cuda()
    rows = fn.affine_sampler_theta_rows(x.shape, Ho)
    out = {}
    try:
        for on in (0, 1):
            _lib.check(L.dspn_affine_sampler_set_batched(on), "set_batched")
            part = torch.full((rows, 6), float("nan"), dtype=torch.float64, device="cuda")
            am = torch.zeros(64, device="cuda")
            dx = fn.affine_sampler_backward_data_theta(dy, th, x, 4, part, dx_absmax=am)
            buf = x.clone()
            part2 = torch.full((rows, 6), float("nan"), dtype=torch.float64, device="cuda")
            inplace = fn.affine_sampler_backward_data_theta(dy, th, buf, 4, part2, dx=buf)
            acc = fn.affine_sampler_backward_data_theta(dy, th, x, 4, part2.clone(), dx=dx.clone(), accumulate=True)
            out[on] = [dx, inplace, acc, part, part2, am.max().reshape(1)]
    finally:
        L.dspn_affine_sampler_set_batched(1)
    assert float(out[1][5]) == float(out[1][0].abs().max())
    exact = tuple(theta) == (1, 0, 0, 0, 1, 0)
    for i, (a, b) in enumerate(zip(out[1], out[0])):
        if exact:
            assert torch.equal(a, b), i
        else:      # generic grids: the source coordinates may differ in the last place between the two kernels (see the kernel's comment)
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-30, i
    # the two entry points run the same kernel: bit for bit, whatever the grid
    assert torch.equal(out[1][0], fn.affine_sampler_backward_data(dy, th, x.shape, 4))
    again = fn.affine_sampler_backward_data_theta(dy, th, x, 4, torch.empty_like(out[1][3]))
    assert torch.equal(again, out[1][0])                                  # and reproducible run to run
    assert L.dspn_affine_sampler_set_batched(2) != 0


@pytest.mark.parametrize("hin,win", [(4, 4), (16, 16), (64, 64), (5, 9)])
def test_affine_sampler_identity_equals_plain_resize(gpu_device, hin, win):
    """with affine_matrix = (1,0,0,0,1,0) the general sampler reproduces dspn_bilinear_forward_f32 bit for bit"""
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, hin, win, 8, generator=g).cuda()
    a = torch.zeros(2, 64, 64, 8, device="cuda")
    b = torch.zeros_like(a)
    fn.bilinear_forward(x, a, 0)
    fn.affine_sampler_forward(fn.SamplerSources([(x, 0)]), torch.tensor([1., 0, 0, 0, 1, 0], device="cuda"), b)
    assert torch.equal(a, b)


def test_softmax_output_valid_normalisation(gpu_device):
    """SoftmaxOutput(multi_output, use_ignore, ignore_label=-1, normalization='valid')"""
    g = torch.Generator().manual_seed(8)
    rows, C = 5000, 9
    logits = (torch.randn(rows, C, generator=g, dtype=torch.float64) * 3).requires_grad_()
    label = torch.randint(-1, C, (rows,), generator=g).double()
    valid = label != -1
    p_ref = torch.softmax(logits, dim=1)
    loss = -(torch.log(p_ref[valid, label[valid].long()])).sum() / valid.sum()
    loss.backward()
    ld = logits.detach().float().cuda()
    lab = label.float().cuda()
    cnt = fn.count(lab, "ne", -1.0)
    assert float(cnt) == float(valid.sum())
    prob, grad = fn.softmax_output(ld, lab, C, -1.0, 1.0, cnt)
    close(prob.cpu().double(), p_ref.detach(), 1e-6)
    close(grad.cpu().double(), logits.grad, 1e-5)
    ce = fn.cross_entropy_sum(prob, lab, C, -1.0, 1e-8).cpu()
    assert float(ce[1]) == float(valid.sum())
    metric = -(torch.log(p_ref.detach().float().double()[valid, label[valid].long()] + 1e-8)).sum() / valid.sum()
    np.testing.assert_allclose(float(ce[0]) / float(ce[1]), float(metric), rtol=1e-5)   # train/metric.py:38-41


def test_softmax_output_seg_padded_channels(gpu_device):
    """seg head: 19 classes in 20 physical channels, ignore 255, grad_scale 4, no normalisation"""
    g = torch.Generator().manual_seed(10)
    rows, C = 4096, 19
    logits = torch.randn(rows, C, generator=g, dtype=torch.float64).requires_grad_()
    label = torch.randint(0, C, (rows,), generator=g).double()
    label[torch.rand(rows, generator=g) < 0.1] = 255
    valid = label != 255
    p_ref = torch.softmax(logits, dim=1)
    (-(torch.log(p_ref[valid, label[valid].long()])).sum() * 4).backward()
    ld = torch.zeros(rows, 20); ld[:, :19] = logits.detach().float(); ld[:, 19] = 7.0
    prob, grad = fn.softmax_output(ld.cuda(), label.float().cuda(), C, 255.0, 4.0, None)
    close(prob[:, :19].cpu().double(), p_ref.detach(), 1e-6)
    close(grad[:, :19].cpu().double(), logits.grad, 1e-5)
    assert float(prob[:, 19].abs().max()) == 0 and float(grad[:, 19].abs().max()) == 0


def test_smooth_l1_makeloss_valid(gpu_device):
    g = torch.Generator().manual_seed(11)
    n = 30000
    pred = (torch.randn(n, generator=g, dtype=torch.float64) * 2).requires_grad_()
    target = torch.randn(n, generator=g, dtype=torch.float64)
    mask = (torch.rand(n, generator=g) < 0.2).double()
    x = mask * (pred - target)
    loss_ref = torch.where(x.abs() < 1, 0.5 * x * x, x.abs() - 0.5)
    nvalid = (loss_ref > 0).sum().clamp(min=1)
    (loss_ref.sum() / nvalid).backward()
    pd, td, md = pred.detach().float().cuda(), target.float().cuda(), mask.float().cuda()
    loss = fn.smooth_l1_forward(pd, td, md)
    close(loss.cpu().double(), loss_ref.detach(), 1e-6)
    cnt = fn.count(loss, "gt", 0.0)
    assert abs(float(cnt) - float(nvalid)) <= 2
    grad = fn.smooth_l1_backward(pd, td, md, cnt)
    close(grad.cpu().double(), pred.grad, 1e-4)
    np.testing.assert_allclose(float(fn.sum_all(loss)), float(loss_ref.sum()), rtol=1e-5)


def test_sgd_momentum_matches_mxnet_rule(gpu_device):
    g = torch.Generator().manual_seed(12)
    n = 4096
    w = torch.randn(n, generator=g, dtype=torch.float64); grad = torch.randn(n, generator=g, dtype=torch.float64)
    mom = torch.randn(n, generator=g, dtype=torch.float64)
    lr, mu, wd, rs = 0.0005, 0.9, 0.0005, 1 / 32
    m_ref = mu * mom - lr * (rs * grad + wd * w)
    w_ref = w + m_ref
    wd_, gd, md = w.float().cuda(), grad.float().cuda(), mom.float().cuda()
    fn.sgd_momentum(wd_, gd, md, lr, mu, wd, rs)
    close(wd_.cpu().double(), w_ref, 1e-6)
    close(md.cpu().double(), m_ref, 1e-6)


def test_layout_helpers(gpu_device):
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 3, 6, 5, generator=g)
    xd = fn.nchw_to_nhwc(x.cuda())
    assert xd.shape == (2, 6, 5, 4)
    assert torch.equal(xd[..., :3].cpu(), x.permute(0, 2, 3, 1)) and float(xd[..., 3].abs().max()) == 0
    assert torch.equal(fn.nhwc_to_nchw(xd, 3).cpu(), x)
    t = torch.randn(3, 7, 9, generator=g)
    assert torch.equal(fn.transpose_bnc(t.cuda()).cpu(), t.permute(0, 2, 1).contiguous())
    a, b = torch.randn(1001, generator=g), torch.randn(1001, generator=g)
    assert torch.equal(fn.add(a.cuda(), b.cuda()).cpu(), a + b)
    # head packing: (B, H*W, 32-padded 30 channels) -> packed (B, total) at an offset, and back
    src = torch.randn(2, 12, 32, generator=g)
    dst = torch.zeros(2, 500)
    out = fn.copy_block(src.cuda(), dst.cuda(), 2, 12, 30, 12 * 32, 32, 0, 500, 30, 100)
    exp = dst.clone(); exp[:, 100:100 + 360] = src[:, :, :30].reshape(2, 360)
    assert torch.equal(out.cpu(), exp)
    y = torch.randn(64, generator=g).clamp(min=0); dy = torch.randn(64, generator=g)
    assert torch.equal(fn.relu_backward(y.cuda(), dy.cuda()).cpu(), dy * (y > 0))


@pytest.mark.parametrize("case", [(2, 32, 32, 3, 64, 7, 2, 3, 1), (3, 9, 11, 3, 16, 3, 1, 1, 1), (2, 12, 12, 4, 8, 3, 2, 0, 1)])
def test_conv_input_sum_grad(gpu_device, case):
    """sum over pixels of the data gradient without forming it (bn_data beta gradient through conv0)"""
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, stride=stride, padding=pad, dilation=dil)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    got = fn.conv2d_input_sum_grad(nhwc(dy), wdev(w), (N, H, W, fn.pad4(Cin)), stride, pad, dil)
    close(got[:Cin].cpu().double(), x.grad.sum(dim=(0, 2, 3)), 1e-4)


@pytest.mark.parametrize("math", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("cin", [32, 48])
@pytest.mark.parametrize("kh,kw,ph,pw,stride", [(1, 7, 0, 3, 1), (7, 1, 3, 0, 1), (1, 3, 0, 1, 1), (3, 1, 1, 0, 1),
                                                (5, 5, 2, 2, 1), (3, 3, 0, 0, 2)])
def test_conv_inception_kernel_classes(gpu_device, kh, kw, ph, pw, stride, cin, math):
    """asymmetric kernels / pads of symbol/inceptionv3.py (1x7, 7x1, 1x3, 3x1, 5x5 p2, 3x3 s2 p0), with a channel count
    whose 32-channel k-steps stay inside one tap (32) and one where they straddle taps (48: the 5x5 tower's input);
    fp32 MFMA, and bf16 MFMA on bf16-representable operands (every product exact, fp32 accumulate: same bound)"""
    g = torch.Generator().manual_seed(kh * 10 + kw)
    N, H, W, Cin, Cout = 2, 17, 17, cin, 48
    rb = (lambda t: t.float().bfloat16().double()) if math == "bf16" else (lambda t: t)
    x = rb(torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)).requires_grad_()
    w = rb(torch.randn(Cout, Cin, kh, kw, generator=g, dtype=torch.float64) / np.sqrt(Cin * kh * kw)).requires_grad_()
    y_ref = F.conv2d(x, w, stride=stride, padding=(ph, pw))
    dy = rb(torch.randn(y_ref.shape, generator=g, dtype=torch.float64))
    y_ref.backward(dy)
    xd, wd_, dyd = nhwc(x.detach()), wdev(w.detach()), nhwc(dy)
    fn.set_conv_math(math)
    try:
        y = fn.conv2d_forward(xd, wd_, None, stride=stride, pad=(ph, pw))
        dx = fn.conv2d_dgrad(dyd, fn.weight_transpose(wd_), tuple(xd.shape), stride=stride, pad=(ph, pw))
        dw = fn.conv2d_wgrad(xd, dyd, tuple(wd_.shape), stride=stride, pad=(ph, pw))
    finally:
        fn.set_conv_math(fn.DEFAULT_CONV_MATH)
    close(nchw(y, Cout), y_ref.detach())
    close(nchw(dx, Cin), x.grad)
    close(dw.cpu().double().permute(0, 3, 1, 2)[:, :Cin], w.grad)


def test_maxpool_full_convention(gpu_device):
    """pooling_convention='full' (symbol/vgg16_reduced.py:40-42): ceil output size, clipped windows"""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 8, 75, 75, generator=g, dtype=torch.float64, requires_grad=True)
    y_ref = F.max_pool2d(x, 2, 2, ceil_mode=True)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    xd = nhwc(x.detach())
    y = fn.maxpool_forward(xd, 2, 2, 0, out=torch.empty(2, 38, 38, 8, device="cuda"))
    assert torch.equal(nchw(y), y_ref.detach().float().double())
    dx = fn.maxpool_backward(xd, y, nhwc(dy), 2, 2, 0)
    close(nchw(dx), x.grad, 1e-6)
    am = torch.zeros(y.shape, dtype=torch.uint8, device="cuda")
    fn.maxpool_forward(xd, 2, 2, 0, out=torch.empty(2, 38, 38, 8, device="cuda"), argmax=am)
    assert torch.equal(fn.maxpool_backward_argmax(am, nhwc(dy), xd.shape, 2, 2, 0), dx)


@pytest.mark.parametrize("case", [(2, 12, 10, 40, 19, 3, 3, 1, 1), (1, 9, 9, 64, 5, 1, 7, 0, 3), (2, 64, 64, 256, 19, 3, 3, 1, 1)])
def test_tap_expanded_conv_matches_direct(gpu_device, conv_math, case):
    """1x1 convolution to Cout*R*S channels + tap_sum == the RxS convolution; tap_spread + 1x1 wgrad == its
    weight gradient (score3_conv's evaluation, engine.Conv(tap_expand=True))"""
    N, H, W, Cin, Cout, R, S, ph, pw = case
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, R, S, generator=g, dtype=torch.float64) * 0.1).requires_grad_()
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    y_ref = F.conv2d(x, w, b, 1, (ph, pw))
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    y_ref.backward(dy)
    xd, wd_, bd = nhwc(x.detach()), wdev(w.detach()), b.float().cuda()
    ldz = fn.pad4(Cout * R * S)
    z = fn.conv2d_forward(xd, wd_.view(Cout * R * S, 1, 1, wd_.shape[3]), None, 1, 0, 1,
                          out=torch.empty(N, H, W, ldz, device="cuda"))
    y = fn.tap_sum(z, bd, Cout, R, S, (ph, pw), out=torch.full((N, H, W, fn.pad4(Cout)), 7.0, device="cuda"))
    close(nchw(y, Cout), y_ref.detach())
    assert float(y[..., Cout:].abs().sum()) == 0.0          # pad channels are written as zeros
    dz = fn.tap_spread(nhwc(dy), Cout, R, S, (ph, pw), out=torch.full((N, H, W, ldz), 7.0, device="cuda"))
    assert float(dz[..., Cout * R * S:].abs().sum()) == 0.0
    dw = fn.conv2d_wgrad(xd, dz, (Cout * R * S, 1, 1, wd_.shape[3]), 1, 0, 1).view(Cout, R, S, -1)
    close(dw[..., :Cin].cpu().double().permute(0, 3, 1, 2), w.grad)


@pytest.fixture
def bf16_math():
    fn.set_conv_math("bf16")
    yield
    fn.set_conv_math(fn.DEFAULT_CONV_MATH)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bf16_mfma_math(gpu_device, bf16_math, case):
    """dspn_conv2d_set_math(1): bf16 MFMA, fp32 accumulate (BASELINE.json configs[3]).  With operands that are exactly
    representable in bf16 every product is exact in fp32, so the result must agree with the float64 reference as
    tightly as the fp32 path does; with general fp32 operands the error is the bf16 rounding of the inputs
    (2^-9 relative per operand)."""
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    rb = lambda t: t.float().bfloat16().double()  # noqa: E731   round to bf16, keep as float64
    x = rb(torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)).requires_grad_()
    w = rb(torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)).requires_grad_()
    y_ref = F.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil)
    dy = rb(torch.randn(y_ref.shape, generator=g, dtype=torch.float64))
    y_ref.backward(dy)
    assert fn.get_conv_math() == "bf16"
    xd, wd_, dyd = nhwc(x.detach()), wdev(w.detach()), nhwc(dy)
    close(nchw(fn.conv2d_forward(xd, wd_, None, stride=stride, pad=pad, dil=dil), Cout), y_ref.detach(), 1e-5)
    if stride == 1 or dil == 1:
        dx = fn.conv2d_dgrad(dyd, fn.weight_transpose(wd_), tuple(xd.shape), stride=stride, pad=pad, dil=dil)
        close(nchw(dx, Cin), x.grad, 1e-5)
    dw = fn.conv2d_wgrad(xd, dyd, tuple(wd_.shape), stride=stride, pad=pad, dil=dil)
    close(dw.cpu().double().permute(0, 3, 1, 2)[:, :Cin], w.grad, 1e-5)
    # general fp32 operands: bounded by the input rounding
    x2 = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    y2 = fn.conv2d_forward(nhwc(x2), wd_, None, stride=stride, pad=pad, dil=dil)
    close(nchw(y2, Cout), F.conv2d(x2, w.detach(), None, stride=stride, padding=pad, dilation=dil), 2e-2)


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1, 1, 1), (2, 17, 19, 32, 48, 3, 2, 1, 1), (3, 16, 16, 64, 256, 1, 1, 0, 1),
                                  (2, 16, 16, 128, 256, 1, 2, 0, 1), (1, 9, 9, 36, 40, 3, 1, 1, 1), (4, 64, 64, 64, 128, 3, 1, 1, 1),
                                  (8, 64, 64, 64, 128, 3, 1, 1, 1), (4, 64, 64, 64, 192, 3, 1, 1, 1)])   # (8 waves; 128x64 tiles)
@pytest.mark.parametrize("relu", [True, False])
def test_conv_with_input_affine(gpu_device, conv_math, case, relu):
    """dspn_conv2d_forward_bn_f32 / dspn_conv2d_wgrad_bn_f32: BatchNorm-apply (+ReLU) folded into the tile loader ==
    the convolution of the materialised (relu)(x*scale+shift), including zero padding AFTER the affine"""
    N, H, W, Cin, Cout, k, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    sc = torch.rand(Cin, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(Cin, generator=g, dtype=torch.float64)
    u = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    u = u.clamp(min=0) if relu else u
    w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k))
    y_ref = F.conv2d(u, w, None, stride=stride, padding=pad, dilation=dil)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
    dw_ref = torch.nn.grad.conv2d_weight(u, w.shape, dy, stride=stride, padding=pad, dilation=dil)
    aff = (sc.float().cuda(), sh.float().cuda(), relu)
    cp = fn.pad4(Cin)
    if cp != Cin:   # pad channels: scale/shift padded with zeros -> u == 0 there
        aff = (torch.cat([aff[0], torch.zeros(cp - Cin, device="cuda")]), torch.cat([aff[1], torch.zeros(cp - Cin, device="cuda")]), relu)
    xd, wd_ = nhwc(x), wdev(w)
    y = fn.conv2d_forward(xd, wd_, None, stride=stride, pad=pad, dil=dil, in_affine=aff)
    close(nchw(y, Cout), y_ref)
    dw = fn.conv2d_wgrad(xd, nhwc(dy), tuple(wd_.shape), stride=stride, pad=pad, dil=dil, in_affine=aff)
    close(dw.cpu().double().permute(0, 3, 1, 2)[:, :Cin], dw_ref)


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1, 1), (3, 16, 16, 64, 256, 1, 1, 0), (2, 17, 19, 32, 48, 3, 2, 1),
                                  (4, 64, 64, 64, 128, 3, 1, 1), (1, 5, 7, 128, 132, 1, 1, 0), (32, 32, 32, 64, 256, 1, 1, 0),
                                  (8, 64, 64, 64, 128, 3, 1, 1), (4, 64, 64, 64, 192, 3, 1, 1),   # 8 waves; 128x64 tiles
                                  (8, 128, 128, 16, 64, 1, 1, 0)])      # last: 2048 tiles -> grouped pre-reduction
@pytest.mark.parametrize("with_res", [False, True])
def test_conv_epilogue_batchnorm_statistics(gpu_device, conv_math, case, with_res):
    """out_stats of dspn_conv2d_forward_bn_f32 + dspn_bn_stats_from_tiles_f32 == dspn_bn_stats_f32 on the stored
    output (mean / rstd / scale / shift), including a large common offset (cancellation) and the residual add"""
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g, dtype=torch.float64) * 30.0          # |mean| >> std
    xd, wd_, bd = nhwc(x), wdev(w), b.float().cuda()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(N, Ho, Wo, Cout, generator=g).cuda() if with_res else None
    tiles, tile_rows = fn.conv_stats_layout(N * Ho * Wo, Cout)
    assert tiles > 0
    st = torch.full((tiles, 2, Cout), float("nan"), device="cuda")
    mm = torch.full((tiles, 2, Cout), float("nan"), device="cuda") if conv_math == "f16x2" else None
    y = fn.conv2d_forward(xd, wd_, bd, stride=stride, pad=pad, residual=res, out_stats=st, out_minmax=mm)
    if mm is not None:
        # two-piece math: per-tile extremes of the stored output beside the statistics.  Exact (they are selections), and
        # the magnitude of a BatchNorm(+ReLU) of y read off the table equals the one taken over the whole tensor
        yt = y.view(-1, Cout)
        assert torch.equal(mm[:, 0].min(0).values, yt.min(0).values) and torch.equal(mm[:, 1].max(0).values, yt.max(0).values)
        rows = N * Ho * Wo
        for t in (0, tiles - 1):
            blk = yt[t * tile_rows:min(rows, (t + 1) * tile_rows)]
            assert torch.equal(mm[t, 0], blk.min(0).values) and torch.equal(mm[t, 1], blk.max(0).values)
        sc = torch.randn(Cout, generator=g).cuda(); sh = torch.randn(Cout, generator=g).cuda()
        for relu_ in (True, False):
            a = fn.absmax(mm.view(-1, Cout), (sc, sh, relu_)).max()
            b = fn.absmax(y, (sc, sh, relu_)).max()
            assert float(a) == float(b) and float(b) > 0
    y_plain = fn.conv2d_forward(xd, wd_, bd, stride=stride, pad=pad, residual=res)
    # (not bit-equal in general: small grids take the split-K path without out_stats, a different summation order)
    assert float((y - y_plain).abs().max()) <= 1e-5 * float(y_plain.abs().max())
    gamma = (torch.rand(Cout, generator=g) + 0.5).cuda(); beta = torch.randn(Cout, generator=g).cuda()
    outs = [torch.empty(Cout, device="cuda") for _ in range(4)]
    fn.bn_stats_from_tiles(st, tiles, tile_rows, N * Ho * Wo, Cout, 2e-5, gamma, beta, *outs)
    if mm is not None:
        # ... and the finalize kernel takes that magnitude itself, for the affine it has just computed (no pass of its own)
        for relu_ in (True, False):
            outs2 = [torch.empty(Cout, device="cuda") for _ in range(4)]
            am = torch.zeros(fn.ABSMAX_SLOTS, device="cuda")
            fn.bn_stats_from_tiles(st, tiles, tile_rows, N * Ho * Wo, Cout, 2e-5, gamma, beta, *outs2, tile_minmax=mm,
                                   relu=relu_, out_absmax=am)
            assert all(torch.equal(a, b) for a, b in zip(outs, outs2))
            assert float(am.max()) == float(fn.absmax(y, (outs2[2], outs2[3], relu_)).max()) > 0
    yd = y.double().view(-1, Cout)
    mean_ref = yd.mean(0); var_ref = yd.var(0, unbiased=False)
    rstd_ref = 1.0 / torch.sqrt(var_ref + 2e-5)
    assert float((outs[0].double() - mean_ref).abs().max()) <= 1e-6 * float(mean_ref.abs().max())
    assert float((outs[1].double() / rstd_ref - 1).abs().max()) <= 2e-6
    ref = [torch.empty(Cout, device="cuda") for _ in range(4)]
    fn.bn_stats(y, 2e-5, gamma, beta, *ref)                      # the separate-pass kernel
    for a, r in zip(outs, ref):
        assert float((a - r).abs().max()) <= 5e-6 * float(r.abs().max())


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 64, 3, 1, 1), (3, 16, 16, 256, 64, 1, 1, 0), (2, 17, 19, 32, 48, 3, 2, 1),
                                  (2, 16, 16, 128, 256, 1, 2, 0), (4, 64, 64, 128, 64, 3, 1, 1), (32, 32, 32, 256, 64, 1, 1, 0),
                                  (8, 64, 64, 128, 128, 3, 1, 1), (4, 64, 64, 192, 64, 3, 1, 1),   # 8 waves; 64x64 tiles
                                  (8, 128, 128, 64, 16, 1, 1, 0)])      # last: 2048 tiles -> grouped pre-reduction
@pytest.mark.parametrize("accumulate", [False, True])
def test_dgrad_epilogue_batchnorm_backward_sums(gpu_device, conv_math, case, accumulate):
    """dspn_conv2d_dgrad_bn_f32 + dspn_bn_backward_from_sums_f32 == dspn_conv2d_dgrad_f32 + dspn_bn_backward_f32
    (stride 1 and the four parity classes of stride 2, with and without accumulation into dx)"""
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 13)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = torch.randn(N, H, W, Cin, generator=g).cuda()                  # BatchNorm input
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin * k * k)).cuda()
    gamma = (torch.rand(Cin, generator=g) + 0.5).cuda(); beta = torch.randn(Cin, generator=g).cuda()
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    wt = fn.weight_transpose(w)
    base = torch.randn(N, H, W, Cin, generator=g).cuda() if accumulate else torch.zeros(N, H, W, Cin, device="cuda")
    # reference: plain dgrad, then the three-kernel BatchNorm backward
    d_ref = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=base.clone(), accumulate=accumulate)
    dx_ref, dg_ref, db_ref = fn.bn_backward(x, scale, shift, d_ref, mean, rstd, gamma, relu=True)
    tiles = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
    assert tiles > 0
    sums = torch.full((tiles, 2, Cin), float("nan"), device="cuda")
    d = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=base.clone(), accumulate=accumulate,
                        bn_bwd=(x, scale, shift, mean, rstd, True, sums))
    assert float((d - d_ref).abs().max()) <= 1e-5 * float(d_ref.abs().max())
    assert torch.isfinite(sums).all()
    dx, dg, db = fn.bn_backward_from_sums(x, scale, shift, d, mean, rstd, gamma, sums, tiles, relu=True)
    for a, r in ((dx, dx_ref), (dg, dg_ref), (db, db_ref)):
        assert float((a - r).abs().max()) <= 2e-5 * float(r.abs().max()), (float((a - r).abs().max()), float(r.abs().max()))


@pytest.mark.parametrize("case", [(2, 24, 24, 96, 1), (2, 16, 16, 48, 1), (8, 128, 128, 64, 1), (4, 128, 128, 256, 1), (3, 25, 23, 64, 2)])
@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("rider", ["wgrad", "wide wgrad", "nobody"])
def test_parked_finalize_rides_in_the_weight_gradient_with_the_same_bits(gpu_device, case, planes, rider):
    """Round 6: dspn_bn_backward_from_sums split in two -- the finalize PARKED (flag | 2 | 8) and run by extra workgroups in
    front of the next weight-gradient launch on the stream (csrc/bn_final_job.h), or by the apply-only call (flag | 4) itself
    when no weight gradient came by -- against the one-call form: dx, dgamma, dbeta, the bound of dx and its per-channel minimum
    bit for bit (grouped and plain tile tables, float dx and piece planes), and the weight gradient that carried the job unchanged."""
    N, H, W, C, stride = case
    if planes and (fn.get_conv_math() != "f16x2" or C % 32):
        pytest.skip("piece planes: the two-piece math, C % 32 == 0")
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(N, H, W, C, generator=g).cuda()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dyb = torch.randn(N, Ho, Wo, 32, generator=g).cuda()
    wb = (torch.randn(32, 1, 1, C, generator=g) / np.sqrt(C)).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda(); beta = torch.randn(C, generator=g).cuda()
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    xr = x.view(-1, C)
    x_ext = torch.stack([xr.min(0).values, xr.max(0).values]).contiguous()
    tiles = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
    sums = torch.zeros(tiles, 2, C, device="cuda"); am_in = torch.zeros(64, device="cuda")
    d = fn.conv2d_dgrad(dyb, fn.weight_transpose(wb), tuple(x.shape), stride, 0, 1, bn_bwd=(x, scale, shift, mean, rstd, True, sums),
                        bn_dy_absmax=am_in)
    # a weight gradient to ride in (any: the job does not touch its operands): both operands as planes for the wide kernel
    xw = torch.randn(2, 20, 20, 128, generator=g).cuda().abs_(); dyw = torch.randn(2, 20, 20, 128, generator=g).cuda()
    one, zero = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    xa, dya = fn.absmax(xw), fn.absmax(dyw)
    xp, dyp = fn.bn_apply_planes(xw, one, zero, xa), fn.bn_apply_planes(dyw, one, zero, dya)
    wide = rider == "wide wgrad"
    if wide and fn.get_conv_math() != "f16x2":
        pytest.skip("the two-piece math is not this process's default")
    def wgrad():
        if wide:
            return fn.conv2d_wgrad(xp, dyp, (128, 3, 3, 128), 1, 1, 1, x_absmax=xa, dy_absmax=dya, x_planes=True, dy_planes=True)
        return fn.conv2d_wgrad(xw, dyw, (128, 3, 3, 128), 1, 1, 1, x_absmax=xa, dy_absmax=dya)
    dw_alone = wgrad()

    def run(split):
        kw = dict(relu=True, dx=torch.empty_like(x), dgamma=torch.zeros(C, device="cuda"), dbeta=torch.zeros(C, device="cuda"))
        bound, bmin = torch.zeros(64, device="cuda"), torch.full((1,), float("inf"), device="cuda")
        if planes:
            kw.update(dx_absmax=bound, dy_absmax=am_in, x_chan_minmax=x_ext, dx_planes=True, dx_absmin=bmin)
        args = (x, scale, shift, d, mean, rstd, gamma, sums, tiles)
        dw = None
        if not split:
            out = fn.bn_backward_from_sums(*args, **kw)
        else:
            fn.bn_backward_from_sums(*args, phase=1, park=True, **kw)
            if rider != "nobody":
                dw = wgrad()
            out = fn.bn_backward_from_sums(*args, phase=2, **kw)
        return out + (bound, bmin), dw

    ref, _ = run(False)
    got, dw = run(True)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(got[1]).all()) and float(got[1].abs().max()) > 0
    if dw is not None:
        assert torch.equal(dw, dw_alone)
    # nothing stays parked: the next weight gradient is a plain one, the next one-call backward unchanged
    assert torch.equal(wgrad(), dw_alone)
    again, _ = run(False)
    for a, b in zip(ref, again):
        assert torch.equal(a, b)


def _decode_planes(planes, shape, block):
    """fp16 piece planes [rows][C / 32][2][32] (stored in a float32 buffer of `shape`) -> float64 values (h0 + h1) / s with the
    power of two s the kernels derive from the magnitude block (csrc/dspn_pieces.h operand_scale)"""
    C = shape[-1]
    h = planes.view(torch.float16).view(-1, C // 32, 2, 32).double()
    m = float(block[torch.isfinite(block)].max())
    e = 15 - (int(np.floor(np.log2(m))) + 1)          # frexp: m = f 2^e', 0.5 <= f < 1
    e = max(-100, min(100, e))
    return ((h[:, :, 0] + h[:, :, 1]) / 2.0 ** e).reshape(shape), 2.0 ** e


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 96, 3, 1, 1), (2, 24, 24, 128, 64, 1, 1, 0), (2, 25, 23, 64, 64, 3, 2, 1)])
def test_gradient_as_piece_planes_from_batchnorm_backward(gpu_device, case):
    """Round 4: the output gradient a BatchNorm backward hands to the convolution in front of it, written
    as fp16 piece planes of the two-piece math instead of floats.  conv_b's data gradient gathers the BatchNorm-backward sums
    and the largest gradient it stores (bn_dy_absmax); dspn_bn_backward_from_sums_f32(dx_planes) bounds its dx from that
    and the per-channel extremes of x, cuts dx by the bound's power of two and writes the planes; conv_a's data gradient
    and weight gradient read them with DSPN_MATH_DY_PLANES.  Checked: the bound IS a bound and within 2^6 of the true
    maximum; the planes decode to the float dx within the rounding of the cut (2^-22 relative to the bound's scale); and
    both consumers give the result of the float path fed the decoded values and the same block (to one ulp: see below)."""
    N, H, W, Ca, Cb, k, stride, pad = case        # conv_a: Ca -> Cb (k x k, stride, pad) -> BatchNorm(+ReLU) -> conv_b: Cb -> 32 (1x1)
    if fn.get_conv_math() != "f16x2":
        pytest.skip("the two-piece math is not this process's default")
    g = torch.Generator().manual_seed(sum(case) + 5)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xa = torch.randn(N, H, W, Ca, generator=g).cuda()
    wa = (torch.randn(Cb, k, k, Ca, generator=g) / np.sqrt(Ca * k * k)).cuda()
    gamma = (torch.rand(Cb, generator=g) * 1.5 + 0.25).cuda(); beta = (torch.randn(Cb, generator=g) * 0.3).cuda()
    wb = (torch.randn(32, 1, 1, Cb, generator=g) / np.sqrt(Cb)).cuda()
    dyb = torch.randn(N, Ho, Wo, 32, generator=g).cuda()
    # forward of conv_a with the statistics / extremes epilogue, BatchNorm finalize with the per-channel extremes
    tiles, tile_rows = fn.conv_stats_layout(N * Ho * Wo, Cb)
    st = torch.zeros(tiles, 2, Cb, device="cuda"); mm = torch.zeros(tiles, 2, Cb, device="cuda")
    x = fn.conv2d_forward(xa, wa, None, stride, pad, 1, out_stats=st, out_minmax=mm)
    mean, rstd, scale, shift = (torch.zeros(Cb, device="cuda") for _ in range(4))
    am_next = torch.zeros(64, device="cuda"); x_ext = torch.zeros(2, Cb, device="cuda")
    fn.bn_stats_from_tiles(st, tiles, tile_rows, N * Ho * Wo, Cb, 2e-5, gamma, beta, mean, rstd, scale, shift,
                           tile_minmax=mm, relu=True, out_absmax=am_next, out_chan_minmax=x_ext)
    xr = x.view(-1, Cb)
    assert torch.equal(x_ext[0], xr.min(0).values) and torch.equal(x_ext[1], xr.max(0).values)
    # backward of conv_b: data gradient with the BatchNorm-backward sums and the magnitude of what it stores
    bt = fn.conv_dgrad_bn_tiles(tuple(x.shape), 1)
    sums = torch.zeros(bt, 2, Cb, device="cuda"); am_in = torch.zeros(64, device="cuda")
    d = fn.conv2d_dgrad(dyb, fn.weight_transpose(wb), tuple(x.shape), 1, 0, 1, bn_bwd=(x, scale, shift, mean, rstd, True, sums),
                        bn_dy_absmax=am_in)
    assert float(am_in.max()) == float(d.abs().max())
    # the BatchNorm backward: floats (reference) and piece planes
    dx_ref, dg_ref, db_ref = fn.bn_backward_from_sums(x, scale, shift, d, mean, rstd, gamma, sums, bt, relu=True)
    bound = torch.zeros(64, device="cuda")
    pl, dg, db = fn.bn_backward_from_sums(x, scale, shift, d, mean, rstd, gamma, sums, bt, relu=True, dx=torch.empty_like(x),
                                          dx_absmax=bound, dy_absmax=am_in, x_chan_minmax=x_ext, dx_planes=True)
    assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)
    true_max, b = float(dx_ref.abs().max()), float(bound.max())
    print(case, "bound / true maximum of dx: %.2f" % (b / true_max))
    assert true_max <= b <= 64.0 * true_max
    dec, s_dx = _decode_planes(pl, tuple(x.shape), bound)
    assert float((dec - dx_ref.double()).abs().max()) <= 2.0 ** -21 * (2.0 ** 15 / s_dx), "the planes are the cut of dx"
    # consumers: conv_a's data gradient and weight gradient on the planes == on the float tensor of the decoded values, same block
    dec32 = dec.float()
    assert torch.equal(dec32.double(), dec)           # h0 + h1 is a float (22 bits)
    wat = fn.weight_transpose(wa)
    dxa_p = fn.conv2d_dgrad(pl, wat, tuple(xa.shape), stride, pad, 1, dy_absmax=bound, dy_planes=True)
    dxa_f = fn.conv2d_dgrad(dec32, wat, tuple(xa.shape), stride, pad, 1, dy_absmax=bound)
    # (not bit for bit: where |h1| is exactly half an ulp of h0, cutting h0 + h1 again rounds the tie to the OTHER neighbour --
    # the same value in two piece pairs whose dropped h1 g1-sized terms differ: a few outputs in a thousand move by one ulp)
    def same(a, b):
        return float((a - b).abs().max()) <= 2.0 ** -20 * float(b.abs().max())
    assert same(dxa_p, dxa_f) and int((dxa_p != dxa_f).sum()) < dxa_f.numel() // 20
    dwa_p = fn.conv2d_wgrad(xa, pl, tuple(wa.shape), stride, pad, 1, dy_absmax=bound, dy_planes=True)
    dwa_f = fn.conv2d_wgrad(xa, dec32, tuple(wa.shape), stride, pad, 1, dy_absmax=bound)
    assert same(dwa_p, dwa_f)
    sc, sh = (torch.rand(Ca, generator=g) + 0.5).cuda(), torch.randn(Ca, generator=g).cuda()     # ... and with a folded input affine
    am_x = fn.absmax(xa, (sc, sh, True))
    dwa_p = fn.conv2d_wgrad(xa, pl, tuple(wa.shape), stride, pad, 1, in_affine=(sc, sh, True), x_absmax=am_x, dy_absmax=bound, dy_planes=True)
    dwa_f = fn.conv2d_wgrad(xa, dec32, tuple(wa.shape), stride, pad, 1, in_affine=(sc, sh, True), x_absmax=am_x, dy_absmax=bound)
    assert same(dwa_p, dwa_f)
    # and against the exact float gradient
    ref = fn.conv2d_dgrad(dx_ref, wat, tuple(xa.shape), stride, pad, 1)
    assert float((dxa_p - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 96, 3, 1, 1), (3, 25, 23, 96, 64, 3, 2, 1), (2, 20, 20, 32, 160, 5, 1, 2),
                                  (1, 17, 17, 128, 192, (1, 7), 1, (0, 3))])
def test_input_as_piece_planes_from_batchnorm_apply(gpu_device, case):
    """Round 4: relu(x * scale + shift) of a BatchNorm in front of a multi-tap convolution written ONCE as
    fp16 piece planes (dspn_bn_apply_planes_f32), cut by the magnitude the statistics finalize formed from the producer's
    extremes; the convolution's forward (with its own statistics epilogue) and weight gradient read them with
    DSPN_MATH_X_PLANES.  The planes hold exactly the pieces the folded-affine loader cuts, so both paths give the same bits;
    the weight gradient is also checked with dy as piece planes (both operands copied)."""
    N, H, W, Cin, Cout, k, stride, pad = case
    if fn.get_conv_math() != "f16x2":
        pytest.skip("the two-piece math is not this process's default")
    kh, kw = fn._hw(k); ph, pw = fn._hw(pad)
    g = torch.Generator().manual_seed(N + H + Cin + Cout + 11)
    x0 = torch.randn(N, H, W, 32, generator=g).cuda()
    w0 = (torch.randn(Cin, 1, 1, 32, generator=g) / np.sqrt(32)).cuda()
    gamma = (torch.rand(Cin, generator=g) * 1.5 + 0.25).cuda(); beta = (torch.randn(Cin, generator=g) * 0.3).cuda()
    w = (torch.randn(Cout, kh, kw, Cin, generator=g) / np.sqrt(Cin * kh * kw)).cuda()
    tiles, tile_rows = fn.conv_stats_layout(N * H * W, Cin)
    st = torch.zeros(tiles, 2, Cin, device="cuda"); mm = torch.zeros(tiles, 2, Cin, device="cuda")
    x = fn.conv2d_forward(x0, w0, None, 1, 0, 1, out_stats=st, out_minmax=mm)
    mean, rstd, scale, shift = (torch.zeros(Cin, device="cuda") for _ in range(4))
    am = torch.zeros(64, device="cuda")
    fn.bn_stats_from_tiles(st, tiles, tile_rows, N * H * W, Cin, 2e-5, gamma, beta, mean, rstd, scale, shift,
                           tile_minmax=mm, relu=True, out_absmax=am)
    act = fn.bn_apply(x, scale, shift, relu=True)
    assert float(am.max()) == float(act.abs().max()), "the finalize's magnitude is that of the activation"
    pl = fn.bn_apply_planes(x, scale, shift, am, relu=True)
    dec, s_x = _decode_planes(pl, tuple(x.shape), am)
    assert float((dec - act.double()).abs().max()) <= 2.0 ** -21 * (2.0 ** 15 / s_x), "the planes are the cut of the activation"
    # forward, with the statistics / extremes epilogue a following BatchNorm asks for
    Ho, Wo = (H + 2 * ph - kh) // stride + 1, (W + 2 * pw - kw) // stride + 1
    t2, _ = fn.conv_stats_layout(N * Ho * Wo, Cout)
    outs = []
    for planes in (False, True):
        st2 = torch.zeros(t2, 2, Cout, device="cuda"); mm2 = torch.zeros(t2, 2, Cout, device="cuda")
        y = fn.conv2d_forward(pl if planes else x, w, None, stride, pad, 1, in_affine=None if planes else (scale, shift, True),
                              out_stats=st2, out_minmax=mm2, x_absmax=am, x_planes=planes)
        y_plain = fn.conv2d_forward(pl if planes else x, w, None, stride, pad, 1, in_affine=None if planes else (scale, shift, True),
                                    x_absmax=am, x_planes=planes)
        # (without the statistics epilogue a small problem may take the split-K path: another summation order)
        assert float((y - y_plain).abs().max()) <= 2e-6 * float(y.abs().max())
        outs.append((y, st2, mm2, y_plain))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b), "planes and folded affine cut the same pieces"
    ref = fn.conv2d_forward(act, w, None, stride, pad, 1, math="fp32")
    assert float((outs[1][0] - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # weight gradient: x planes with a float dy, and with dy as planes too
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).cuda()
    am_dy = fn.absmax(dy)
    dw_f = fn.conv2d_wgrad(x, dy, tuple(w.shape), stride, pad, 1, in_affine=(scale, shift, True), x_absmax=am, dy_absmax=am_dy)
    dw_p = fn.conv2d_wgrad(pl, dy, tuple(w.shape), stride, pad, 1, x_absmax=am, dy_absmax=am_dy, x_planes=True)
    assert torch.equal(dw_f, dw_p)
    ref = fn.conv2d_wgrad(act, dy, tuple(w.shape), stride, pad, 1, math="fp32")
    assert float((dw_p - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    if Cout % 32 == 0:
        one, zero = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
        dyp = fn.bn_apply_planes(dy, one, zero, am_dy)            # (identity affine: dy cut into planes by its own block)
        dw_pp = fn.conv2d_wgrad(pl, dyp, tuple(w.shape), stride, pad, 1, x_absmax=am, dy_absmax=am_dy, x_planes=True, dy_planes=True)
        assert torch.equal(dw_pp, dw_p)
        dw_fp = fn.conv2d_wgrad(x, dyp, tuple(w.shape), stride, pad, 1, in_affine=(scale, shift, True), x_absmax=am, dy_absmax=am_dy,
                                dy_planes=True)
        assert torch.equal(dw_fp, dw_p)
    slabs_n = fn.conv2d_wgrad_splits(tuple(x.shape), tuple(dy.shape), tuple(w.shape), stride)
    if slabs_n > 0:
        sl_f = torch.zeros(slabs_n, w.numel(), device="cuda"); sl_p = torch.zeros_like(sl_f)
        fn.conv2d_wgrad_slabs(x, dy, tuple(w.shape), sl_f, stride, pad, 1, in_affine=(scale, shift, True), x_absmax=am, dy_absmax=am_dy)
        fn.conv2d_wgrad_slabs(pl, dy, tuple(w.shape), sl_p, stride, pad, 1, x_absmax=am, dy_absmax=am_dy, x_planes=True)
        assert torch.equal(sl_f, sl_p)


@pytest.mark.parametrize("rows,C,ld", [(1000, 64, 64), (70000, 30, 32), (5, 8, 8), (300000, 128, 128)])
def test_relu_backward_colsum(gpu_device, rows, C, ld):
    g = torch.Generator().manual_seed(rows + C)
    y = torch.randn(rows, ld, generator=g).clamp(min=0).cuda()
    dy = torch.randn(rows, ld, generator=g).cuda()
    ref_dx = torch.where(y > 0, dy, torch.zeros_like(dy))
    ref_sum = ref_dx.double().sum(0)[:C]
    dx, out = fn.relu_backward_colsum(y, dy.clone(), C, out=torch.empty(C, device="cuda"))
    assert torch.equal(dx, ref_dx)
    assert float((out.double() - ref_sum).abs().max()) <= 1e-5 * float(ref_sum.abs().max() + 1)
    out2 = fn.colsum(ref_dx, C)
    assert float((out2.double() - ref_sum).abs().max()) <= 1e-5 * float(ref_sum.abs().max() + 1)
    # round 5: the magnitude block of dx as a by-product of the same pass
    block = torch.zeros(fn.ABSMAX_SLOTS, device="cuda")
    dx3, out3 = fn.relu_backward_colsum(y, dy.clone(), C, out=torch.empty(C, device="cuda"), dx_absmax=block)
    assert torch.equal(dx3, ref_dx) and torch.equal(out3, out)
    assert float(block.max()) == float(ref_dx.abs().max()) == float(fn.absmax(ref_dx).max())


# BASELINE.json's full sizes (batch 32 at 512x512): where the fp64 CPU reference of the cases above would take minutes,
# the three kernels are tied to each other by size-independent properties and to an independent GPU implementation
FULL_SIZE_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (32, 128, 128, 64, 64, 3, 1, 1),      # stage1 conv2
    (32, 128, 128, 256, 64, 1, 1, 0),     # stage1 conv1
    (32, 128, 128, 256, 128, 3, 2, 1),    # stage2 entry (strided data gradient: 4 parity classes)
    (32, 64, 64, 256, 512, 1, 2, 0),      # stage2 shortcut (3 of 4 parity classes empty)
    (32, 32, 32, 256, 256, 3, 1, 1),      # stage3 conv2
    (32, 16, 16, 512, 512, 3, 1, 1),      # stage4 conv2
    (32, 512, 512, 4, 64, 7, 2, 3),       # conv0 (3 channels padded to 4)
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", FULL_SIZE_CASES)
def test_conv_full_size_properties(gpu_device, conv_math, case):
    """<conv(x; w), y> = <x, dgrad(y; w)> = <w, wgrad(x, y)> (the three kernels are adjoint views of one trilinear
    form), linearity of the forward, and agreement with torch's own ROCm convolution (MIOpen) on the same operands.
    Tolerances: 2e-5 relative on the inner products (fp64 reductions of fp32 results), 1e-4 of the output's max on the
    element-wise comparisons (different summation orders over K up to 4608)."""
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator(device="cuda").manual_seed(sum(case))
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
    if Cin == 4:
        x[..., 3] = 0                                     # the pad channel of conv0's input is zero
    w = torch.randn(Cout, k, k, Cin, device="cuda", generator=g) / np.sqrt(Cin * k * k)
    y = fn.conv2d_forward(x, w, None, stride=stride, pad=pad, dil=1)
    dy = torch.randn(y.shape, device="cuda", generator=g)
    wt = fn.weight_transpose(w)
    dx = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride=stride, pad=pad, dil=1)
    dw = fn.conv2d_wgrad(x, dy, tuple(w.shape), stride=stride, pad=pad, dil=1)
    a = float((y.double() * dy.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((w.double() * dw.double()).sum())
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(a - b) <= 2e-5 * scale and abs(a - c) <= 2e-5 * scale, (a, b, c, scale)
    # linearity in x
    x2 = torch.randn(N, H, W, Cin, device="cuda", generator=g)
    if Cin == 4:
        x2[..., 3] = 0
    y2 = fn.conv2d_forward(x2, w, None, stride=stride, pad=pad, dil=1)
    y12 = fn.conv2d_forward(0.5 * x + x2, w, None, stride=stride, pad=pad, dil=1)
    assert float((y12 - (0.5 * y + y2)).abs().max()) <= 1e-4 * float(y.abs().max())
    # an independent implementation on the same device
    ref = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, stride=stride, padding=pad)
    assert float((y.permute(0, 3, 1, 2) - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    xr = x.permute(0, 3, 1, 2).detach().requires_grad_()
    wr = w.permute(0, 3, 1, 2).detach().requires_grad_()
    F.conv2d(xr, wr, None, stride=stride, padding=pad).backward(dy.permute(0, 3, 1, 2))
    assert float((dx.permute(0, 3, 1, 2) - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())
    assert float((dw.permute(0, 3, 1, 2) - wr.grad).abs().max()) <= 2e-4 * float(wr.grad.abs().max())


@pytest.mark.parametrize("case", [(8, 32, 32, 256, 256, 3, 1, 1), (8, 64, 64, 128, 512, 1, 1, 0), (8, 64, 64, 64, 64, 3, 1, 1),
                                  (4, 64, 64, 256, 128, 1, 2, 0), (2, 32, 32, 36, 40, 3, 1, 1)])
def test_split_math_is_as_accurate_as_the_fp32_mfma(gpu_device, case):
    """DSPN_MATH_F32_F16X2 (two fp16 pieces, three products, the default) and DSPN_MATH_F32_BF16X3 (three bf16 pieces, six
    products) against DSPN_MATH_FP32, all against float64, on operands with full 24-bit mantissas: forward, data gradient and
    weight gradient (plain, and with the fused input affine + output statistics).  A split mode's root-mean-square error
    may not exceed 1.25x the fp32 MFMA's on any of them and its largest error 2x (measured: bf16x3 rms 0.85 .. 1.05x, largest
    0.8 .. 1.6x; f16x2 rms 1.1 .. 1.2x, largest 0.6 .. 0.9x), and all stay below 1e-5 of the tensor's scale -- i.e. both are
    fp32 convolutions, not reduced-precision ones."""
    N, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64).float().double().requires_grad_()
    w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)).float().double().requires_grad_()
    sc = (torch.rand(Cin, generator=g, dtype=torch.float64) + 0.5).float().double()
    sh = torch.randn(Cin, generator=g, dtype=torch.float64).float().double()
    y_ref = F.conv2d(x, w, None, stride=stride, padding=pad)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64).float().double()
    y_ref.backward(dy)
    u = (x.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).float().double().clamp(min=0)   # the device forms u in fp32 (one fmaf)
    yu_ref = F.conv2d(u, w.detach(), None, stride=stride, padding=pad)
    dwu_ref = torch.nn.grad.conv2d_weight(u, w.shape, dy, stride=stride, padding=pad)
    xd, wd_, dyd = nhwc(x.detach()), wdev(w.detach()), nhwc(dy)
    cp = fn.pad4(Cin)
    aff = (torch.cat([sc.float(), torch.zeros(cp - Cin)]).cuda(), torch.cat([sh.float(), torch.zeros(cp - Cin)]).cuda(), True)

    def rel(got, exp):      # (largest, root-mean-square) error in units of the tensor's largest entry
        d = (got.double().cpu() - exp)
        return float(d.abs().max() / exp.abs().max()), float((d * d).mean().sqrt() / exp.abs().max())

    errs = {}
    for mode in ("fp32", "bf16x3", "f16x2"):
        fn.set_conv_math(mode)
        try:
            y = fn.conv2d_forward(xd, wd_, None, stride=stride, pad=pad)
            dx = fn.conv2d_dgrad(dyd, fn.weight_transpose(wd_), tuple(xd.shape), stride=stride, pad=pad)
            dw = fn.conv2d_wgrad(xd, dyd, tuple(wd_.shape), stride=stride, pad=pad)
            yu = fn.conv2d_forward(xd, wd_, None, stride=stride, pad=pad, in_affine=aff)
            dwu = fn.conv2d_wgrad(xd, dyd, tuple(wd_.shape), stride=stride, pad=pad, in_affine=aff)
        finally:
            fn.set_conv_math(fn.DEFAULT_CONV_MATH)
        errs[mode] = dict(fwd=rel(nchw(y, Cout), y_ref.detach()), dgrad=rel(nchw(dx, Cin), x.grad),
                          wgrad=rel(dw.permute(0, 3, 1, 2)[:, :Cin], w.grad), fwd_affine=rel(nchw(yu, Cout), yu_ref),
                          wgrad_affine=rel(dwu.permute(0, 3, 1, 2)[:, :Cin], dwu_ref))
    for split in ("bf16x3", "f16x2"):
        print(case, split, {k: "max %.2f rms %.2f of fp32's" % (errs[split][k][0] / errs["fp32"][k][0], errs[split][k][1] / errs["fp32"][k][1])
                            for k in errs["fp32"]})
        for key in errs["fp32"]:
            (mx, rx), (mf, rf) = errs[split][key], errs["fp32"][key]
            assert mx < 1e-5 and mf < 1e-5, (split, key, errs)
            # the root-mean-square error is the stable statistic (the largest of 10^5 .. 10^7 errors fluctuates by tens of
            # percent between two evaluations of equal quality): within 1.25x; the largest error within 2x
            assert rx <= 1.25 * rf + 1e-9 and mx <= 2.0 * mf + 1e-8, (split, key, errs)


@pytest.mark.parametrize("xs,ws", [(1.0, 1.0), (1e-20, 1e-12), (1e20, 1e12), (3e-30, 1.0), (1.0, 1e25)])
def test_two_piece_math_scales_any_float_magnitude_into_fp16_range(gpu_device, xs, ws):
    """DSPN_MATH_F32_F16X2: the per-tensor power-of-two scales make the result independent of the operands' magnitudes --
    the same convolution with x scaled by xs and w by ws gives exactly (xs * ws) times the unit-scale result when both are
    powers of two, and the fp32 accuracy otherwise; forward, data gradient and weight gradient."""
    fn.set_conv_math("f16x2")
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4, 32, 32, 64, generator=g).cuda(); w = (torch.randn(96, 3, 3, 64, generator=g) / 24).cuda()
        dy = torch.randn(4, 32, 32, 96, generator=g).cuda()
        y0 = fn.conv2d_forward(x, w, None, 1, 1, 1)
        dx0 = fn.conv2d_dgrad(dy, fn.weight_transpose(w), tuple(x.shape), 1, 1, 1)
        dw0 = fn.conv2d_wgrad(x, dy, tuple(w.shape), 1, 1, 1)
        y = fn.conv2d_forward(x * xs, w * ws, None, 1, 1, 1)
        dx = fn.conv2d_dgrad(dy * xs, fn.weight_transpose(w * ws), tuple(x.shape), 1, 1, 1)
        dw = fn.conv2d_wgrad(x * xs, dy * ws, tuple(w.shape), 1, 1, 1)
        for got, ref in ((y, y0), (dx, dx0), (dw, dw0)):
            assert torch.isfinite(got).all()
            exp = ref.double() * xs * ws
            assert float((got.double() - exp).abs().max()) <= 2e-6 * float(exp.abs().max())
    finally:
        fn.set_conv_math(fn.DEFAULT_CONV_MATH)


@pytest.mark.parametrize("cin", [64, 20])
def test_two_piece_math_propagates_non_finite_operands_like_fp32(gpu_device, cin):
    """Round 4: +-inf and NaN elements of either operand in DSPN_MATH_F32_F16X2 against the fp32 MFMA.
    Everywhere: an output is non-finite exactly where fp32's is, NaN where fp32 has NaN, and the outputs no non-finite
    element touches stay as accurate as before (the scale comes from the FINITE partial maxima of the magnitude block).
    Signed infinities are reproduced where the pieces of an infinite element are repaired to (+-65504, +-inf): weights cut
    by the piece-plane kernel (cin = 64) and both operands of the weight gradient.  conv_nt_kernel's own loaders do not
    repair (include/dspn_nn.h says why): an infinite activation / output gradient -- and, with cin = 20, an infinite weight
    cut inside the kernel -- gives NaN where fp32 gives +-inf."""
    g = torch.Generator().manual_seed(17)
    N, H, W, Cout = 2, 12, 12, 48
    x = torch.randn(N, H, W, cin, generator=g); w = torch.randn(Cout, 3, 3, cin, generator=g) / 12
    dy = torch.randn(N, H, W, Cout, generator=g)
    inf = float("inf")
    x[0, 3, 3, 1] = inf; x[0, 8, 2, 5] = -inf; x[1, 5, 5, 2] = float("nan"); x[1, 9, 9, 7] = inf; x[1, 9, 10, 7] = -inf
    w[5, 1, 1, 1] = 0.0                      # inf * 0 -> NaN at (0, 3, 3) of output channel 5
    dy[0, 6, 6, 3] = inf; dy[1, 2, 9, 11] = float("nan")
    w2 = w.clone(); w2[7, 0, 2, 3] = -inf    # an infinite WEIGHT (forward and data gradient)
    x, w, w2, dy = x.cuda(), w.cuda(), w2.cuda(), dy.cuda()
    res = {}
    for mode in ("fp32", "f16x2"):
        fn.set_conv_math(mode)
        try:
            res[mode] = dict(fwd=fn.conv2d_forward(x, w, None, 1, 1, 1), fwd_w=fn.conv2d_forward(x.nan_to_num(0, 0, 0), w2, None, 1, 1, 1),
                             dgrad=fn.conv2d_dgrad(dy, fn.weight_transpose(w), tuple(x.shape), 1, 1, 1),
                             dgrad_w=fn.conv2d_dgrad(dy.nan_to_num(0, 0, 0), fn.weight_transpose(w2), tuple(x.shape), 1, 1, 1),
                             wgrad=fn.conv2d_wgrad(x, dy, tuple(w.shape), 1, 1, 1))
        finally:
            fn.set_conv_math(fn.DEFAULT_CONV_MATH)
    # dgrad_w reads the TRANSPOSED weight: its contraction runs over Cout = 48 channels -- not a multiple of 32, cut in the kernel
    exact_inf = {"wgrad"} | ({"fwd_w"} if cin % 32 == 0 else set())
    for key, ref in res["fp32"].items():
        got = res["f16x2"][key]
        fin = torch.isfinite(ref)
        assert 0 < int((~fin).sum()) < ref.numel() // 2, key            # the case does exercise both kinds of output
        assert torch.equal(torch.isfinite(got), fin), (key, int((torch.isfinite(got) != fin).sum()))
        assert bool(torch.isnan(got)[torch.isnan(ref)].all()), key        # NaN wherever fp32 has NaN
        if key in exact_inf:
            assert torch.equal(torch.isnan(got), torch.isnan(ref)), key
            assert torch.equal(got[torch.isinf(ref)], ref[torch.isinf(ref)]), key      # the same signed infinities
        scale = float(ref[fin].abs().max())
        assert float((got[fin] - ref[fin]).abs().max()) <= 1e-5 * scale, key


def test_two_piece_math_with_a_wide_dynamic_range_inside_one_tensor(gpu_device):
    """DSPN_MATH_F32_F16X2's documented limit: elements more than 2^17 below their tensor's largest magnitude keep an ABSOLUTE
    error (2^-39 of that maximum) instead of a relative one.  One outlier of 2^20 in an otherwise unit-scale input: the
    outputs it does not touch still agree with float64 to 1e-5 of the unit-scale output (the outlier's own outputs to 2e-6 of
    theirs), where the three-piece bf16 split and the fp32 MFMA give 2e-6 everywhere."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 16, 64, generator=g, dtype=torch.float64).float()
    w = (torch.randn(64, 3, 3, 64, generator=g, dtype=torch.float64) / 24).float()
    x[0, 2, 2, 5] = 2.0 ** 20
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    far = torch.ones(2, 16, 16, dtype=torch.bool); far[0, 1:4, 1:4] = False          # outputs the outlier does not reach
    unit = float(ref[far].abs().max())
    for mode, tol in (("f16x2", 1e-5), ("bf16x3", 2e-6), ("fp32", 2e-6)):
        fn.set_conv_math(mode)
        try:
            y = fn.conv2d_forward(x.cuda(), w.cuda(), None, 1, 1, 1).double().cpu()
        finally:
            fn.set_conv_math(fn.DEFAULT_CONV_MATH)
        assert float((y - ref)[far].abs().max()) <= tol * unit, mode
        assert float((y - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), mode


@pytest.mark.parametrize("k,cout", [(1, 64), (3, 32)])
def test_conv_tensors_of_2_gib_and_more_run_in_batch_chunks(gpu_device, conv_math, k, cout):
    """The buffer-addressed kernels take tensors below 2 GiB (32-bit offsets, bit 31 = out of range); the C entry points cut
    larger batches into chunks of images (conv.hip batch_chunk).  x here is 130 x 128 x 128 x 256 floats = 2.03 GiB:
    forward, data gradient (dx of 2.03 GiB written by chunks of dy) and weight gradient (accumulated over the chunks)
    against torch's own ROCm convolution evaluated on sub-batches of 26 images."""
    N, H, W, Cin = 130, 128, 128, 256
    g = torch.Generator(device="cuda").manual_seed(k + cout)
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
    assert x.numel() * 4 >= 2 ** 31
    w = torch.randn(cout, k, k, Cin, device="cuda", generator=g) / np.sqrt(Cin * k * k)
    dy = torch.randn(N, H, W, cout, device="cuda", generator=g)
    pad = k // 2
    y = fn.conv2d_forward(x, w, None, 1, pad, 1)
    dx = fn.conv2d_dgrad(dy, fn.weight_transpose(w), tuple(x.shape), 1, pad, 1)
    dw = fn.conv2d_wgrad(x, dy, tuple(w.shape), 1, pad, 1)
    wr = w.permute(0, 3, 1, 2).contiguous()
    dw_ref = torch.zeros_like(wr, dtype=torch.float64)
    ymax = dxmax = yerr = dxerr = 0.0
    for n0 in range(0, N, 26):
        xs = x[n0:n0 + 26].permute(0, 3, 1, 2).detach().requires_grad_()
        ws = wr.detach().requires_grad_()
        ys = F.conv2d(xs, ws, None, 1, pad)
        ys.backward(dy[n0:n0 + 26].permute(0, 3, 1, 2))
        yerr = max(yerr, float((y[n0:n0 + 26].permute(0, 3, 1, 2) - ys).abs().max())); ymax = max(ymax, float(ys.abs().max()))
        dxerr = max(dxerr, float((dx[n0:n0 + 26].permute(0, 3, 1, 2) - xs.grad).abs().max())); dxmax = max(dxmax, float(xs.grad.abs().max()))
        dw_ref += ws.grad.double()
        del xs, ys
    assert yerr <= 1e-4 * ymax and dxerr <= 1e-4 * dxmax, (yerr, ymax, dxerr, dxmax)
    dwerr = float((dw.permute(0, 3, 1, 2).double() - dw_ref).abs().max())
    assert dwerr <= 2e-4 * float(dw_ref.abs().max()), (dwerr, float(dw_ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c,soff,doff", [(64, 0, 96), (32, 16, 8), (24, 8, 0), (20, 4, 12)])
def test_copy_block_in_16_byte_units_and_by_element(gpu_device, dtype, c, soff, doff):
    """the channel-slice copy behind Concat / the SSD head packing: whole 16-byte units when every extent, offset and stride
    allows (4 floats / 8 bf16), element by element otherwise -- both against torch slicing, forward (into a slice) and
    backward (out of a slice), incl. untouched neighbours"""
    N, H, W, Cs, Cd = 3, 5, 7, 96, 160
    g = torch.Generator().manual_seed(c + soff + doff)
    src = torch.randn(N, H, W, Cs, generator=g).to(dtype).cuda()
    dst = torch.randn(N, H, W, Cd, generator=g).to(dtype).cuda()
    ref = dst.clone(); ref[..., doff:doff + c] = src[..., soff:soff + c]
    fn.copy_block(src, dst, N, H * W, c, H * W * Cs, Cs, soff, H * W * Cd, Cd, doff)
    assert torch.equal(dst, ref)
    back = torch.randn(N, H, W, Cs, generator=g).to(dtype).cuda()
    ref2 = back.clone(); ref2[..., soff:soff + c] = dst[..., doff:doff + c]
    fn.copy_block(dst, back, N, H * W, c, H * W * Cd, Cd, doff, H * W * Cs, Cs, soff)
    assert torch.equal(back, ref2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k,stride,pad", [((2, 32, 32, 64), 3, 2, 1), ((3, 17, 23, 8), 3, 2, 1), ((2, 16, 16, 32), 2, 2, 0),
                                                ((1, 12, 12, 16), 3, 1, 1)])
def test_bn_backward_with_the_pooled_gradient_formed_on_the_fly(gpu_device, shape, k, stride, pad):
    """dspn_bn_backward_maxpool_f32 (round 4; the resnet stem bn0 -> relu0 -> pooling0, symbol/resnet.py:96-98): the BatchNorm
    backward that gathers its output gradient from (pooled gradient, argmax record) gives what the two separate operators
    give (the reductions bit for bit, dx to a rounding) -- dspn_maxpool_backward_argmax_f32 into a dense tensor, then dspn_bn_backward_f32 on it"""
    N, H, W, C = shape
    g = torch.Generator().manual_seed(sum(shape) + k)
    x = torch.randn(N, H, W, C, generator=g).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda(); beta = (torch.randn(C, generator=g) * 0.2).cuda()
    mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(N, Ho, Wo, C, device="cuda"); argmax = torch.zeros(N, Ho, Wo, C, dtype=torch.uint8, device="cuda")
    fn.maxpool_forward(x, k, stride, pad, out=y, argmax=argmax, in_affine=(scale, shift, True))
    dyp = torch.randn(N, Ho, Wo, C, generator=g).cuda()
    dense = fn.maxpool_backward_argmax(argmax, dyp, (N, H, W, C), k, stride, pad)
    am_ref, am = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    dx_ref, dg_ref, db_ref = fn.bn_backward(x, scale, shift, dense, mean, rstd, gamma, relu=True, dx_absmax=am_ref)
    dx, dg, db = fn.bn_backward_maxpool(x, scale, shift, dyp, argmax, k, stride, pad, mean, rstd, gamma, relu=True, dx_absmax=am)
    assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)
    # (dx = a dy' + c1 x + c0 in two kernels: the same three terms, possibly contracted into fused multiply-adds differently)
    assert float((dx - dx_ref).abs().max()) <= 2.0 ** -22 * float(dx_ref.abs().max())
    assert float(am.max()) == float(dx.abs().max()) and float(am_ref.max()) == float(dx_ref.abs().max())


@pytest.mark.gpu
def test_copy_block_batch_packs_head_maps_like_the_single_copies(gpu_device):
    """dspn_copy_block_batch_f32 (round 4): the SSD head packing of one pass -- six maps of different sizes, channel padding
    stripped, each at its offset of the (B, total) row -- as one launch, and its gradient (out of the row, into the padded
    maps, with and without accumulation): bit for bit what the per-map dspn_copy_block_f32 calls give"""
    g = torch.Generator().manual_seed(4)
    B, maps = 3, [(16, 16, 20, 20), (8, 8, 30, 32), (4, 4, 30, 32), (2, 2, 30, 32), (1, 1, 20, 20), (5, 3, 54, 56)]   # H, W, c, ldc
    srcs = [torch.randn(B, H, W, ld, generator=g).cuda() for (H, W, c, ld) in maps]
    sizes = [H * W * c for (H, W, c, ld) in maps]
    offs = np.cumsum([0] + sizes).tolist(); total = offs[-1]
    out = torch.full((B, total), float("nan"), device="cuda"); ref = out.clone()
    entries = [(t, out, B, H * W, c, H * W * ld, ld, 0, total, c, off, False) for t, (H, W, c, ld), off in zip(srcs, maps, offs)]
    fn.copy_block_batch(*fn.copy_block_table(entries, out.device))
    for e in entries:
        fn.copy_block(e[0], ref, *e[2:-1])
    assert torch.equal(out, ref) and bool(torch.isfinite(out).all())
    for t, (H, W, c, ld), off in zip(srcs, maps, offs):
        assert torch.equal(out[:, off:off + H * W * c].view(B, H, W, c), t[..., :c])
    grad = torch.randn(B, total, generator=g).cuda()
    dxs = [torch.randn(B, H, W, ld, generator=g).cuda() for (H, W, c, ld) in maps]
    refs = [d.clone() for d in dxs]
    accs = [False, True, False, True, True, False]
    entries = [(grad, d, B, H * W, c, total, c, off, H * W * ld, ld, 0, a) for d, (H, W, c, ld), off, a in zip(dxs, maps, offs, accs)]
    fn.copy_block_batch(*fn.copy_block_table(entries, grad.device))
    for e, r in zip(entries, refs):
        fn.copy_block(e[0], r, *e[2:-1], accumulate=e[-1])
    for d, r in zip(dxs, refs):
        assert torch.equal(d, r)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [False, True])
def test_batched_weight_transposes_match_permute(gpu_device, half):
    """dspn_conv2d_weight_transpose_batch_f32 / dspn_conv2d_weight_prepare_batch_bf16 (one 32 x 32 tile of a tap per
    workgroup): wt[c][t][k] = w[k][t][c] zero padded to Kp, and the storage-type copy of w itself in the bf16 form -- shapes
    with partial tiles in both directions, several rows in one table"""
    dt = torch.bfloat16 if half else torch.float32
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 3, 3, 64, 64), (19, 1, 1, 128, 24), (40, 3, 3, 36, 40), (171, 1, 1, 2048, 176), (96, 1, 7, 160, 96), (8, 7, 7, 4, 8)]
    trip = []
    for (K, R, S, C, Kp) in shapes:
        w = torch.randn(K, R, S, C, generator=g).cuda()
        wt = torch.full((C, R, S, Kp), float("nan"), device="cuda").to(dt)
        wh = torch.full((K, R, S, C), float("nan"), device="cuda").to(dt) if half else None
        trip.append((w, wt, wh))
    fn.weight_transpose_batch(*fn.weight_transpose_table(trip, torch.device("cuda")))
    for (w, wt, wh), (K, R, S, C, Kp) in zip(trip, shapes):
        ref = torch.zeros(C, R, S, Kp, device="cuda")
        ref[..., :K] = w.permute(3, 1, 2, 0)
        assert torch.equal(wt, ref.to(dt))
        if half:
            assert torch.equal(wh, w.to(dt))


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(4, 64, 64, 64, 128, 3, 1, 1, True),      # float A operand on the wide family (conv_ntv_kernel)
                                  (4, 64, 64, 256, 64, 1, 1, 0, False),     # 64 output columns: the 256 x 64 tile
                                  (2, 9, 9, 32, 24, 3, 2, 1, True),         # small: conv_nt_kernel, followed by the pass over out
                                  (1, 5, 7, 8, 12, 3, 1, 1, False)])
def test_conv_forward_leaves_the_magnitude_of_its_output(gpu_device, case):
    """round 5: fn.conv2d_forward(out_absmax=block) -- a convolution no BatchNorm reads (vgg16_reduced, the SSD extra layers)
    leaves the magnitude block of what it stored (after bias and ReLU), which is what fn.absmax(out) would give, so that the
    next convolution of the two-piece math needs no pass over the tensor"""
    N, H, W, Cin, Cout, k, stride, pad, relu = case
    g = torch.Generator(device="cuda").manual_seed(sum(case[:8]))
    x = torch.randn(N, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, k, k, Cin, device="cuda", generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, device="cuda", generator=g)
    block = torch.zeros(fn.ABSMAX_SLOTS, device="cuda")
    y = fn.conv2d_forward(x, w, b, stride=stride, pad=pad, dil=1, relu=relu, math="f16x2", out_absmax=block)
    ref = fn.conv2d_forward(x, w, b, stride=stride, pad=pad, dil=1, relu=relu, math="f16x2")
    assert torch.equal(y, ref)                                  # the epilogue that tracks the magnitude stores the same bits
    assert float(block.max()) == float(y.abs().max()) > 0
    assert float(fn.absmax(y).max()) == float(block.max())
    # a second call accumulates by maximum (the caller zeroes the block once per step)
    fn.conv2d_forward(0.5 * x, w, None, stride=stride, pad=pad, dil=1, relu=relu, math="f16x2", out_absmax=block)
    assert float(block.max()) == float(y.abs().max())
