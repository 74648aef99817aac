"""Round 6: one MEASURED Winograd F(2x2, 3x3) prototype for the stage-3 conv2 forward of resnet-50 (32 images,
32 x 32 x 256 -> 256, 3 x 3 / 1), in the two-piece fp16 math, with the scale taken AFTER the input transform.

What is measured on the device, with the library's own kernels:
  direct      the layer as the graph runs it (x as fp16 piece planes, conv_ntw_kernel), time and error against float64;
  gemm/planes the 16 xi-GEMMs of the Winograd form (8192 tiles x 256 -> 256 each) as ONE plane-fed 1 x 1 launch over all
              16 x 8192 rows -- what a grouped launch of them costs when the transformed input is ALREADY piece planes in memory
              (an unfused pipeline: the input transform writes 4 x the tensor, the output transform reads 4 x the output);
  gemm/float  the same GEMM with the A operand cut in the loader (conv_ntv_kernel) -- the loader a FUSED kernel needs (the
              transformed tile exists only in registers, so it cannot come by LDS-DMA), still without the transform's additions;
and numerically (device GEMMs per xi with their own scale, transforms in fp32): the error of the Winograd result against float64
next to the direct kernel's and the fp32 MFMA's.
Adoption bar (VERDICT): >= 15 % faster per layer and <= 2 x the fp32 MFMA's error."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
N, H, W, C, K = 32, 32, 32, 256, 256


def timeit(f, reps=10):
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


g = torch.Generator().manual_seed(6)
x = torch.randn(N, H, W, C, generator=g).clamp_(min=0) * 1.3          # a BatchNorm + ReLU output
w = torch.randn(K, 3, 3, C, generator=g) / np.sqrt(9 * C)
ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
xd, wd = x.to(dev), w.to(dev)
scale = float(ref.abs().max())

# ---- direct, as the graph runs it
xp, xa = planes_of(xd)
wa = fn.absmax(wd); wp = fn.weight_planes(wd, math="f16x2", w_absmax=wa)
y = torch.empty(N, H, W, K, device=dev)
direct = lambda: fn.conv2d_forward(xp, wd, None, 1, 1, 1, w_planes=wp, x_absmax=xa, w_absmax=wa, x_planes=True, out=y)
t_direct = timeit(direct)
err_direct = float((y.double().cpu() - ref).abs().max()) / scale
y32 = fn.conv2d_forward(xd, wd, None, 1, 1, 1, math="fp32")
err_fp32 = float((y32.double().cpu() - ref).abs().max()) / scale

# ---- Winograd F(2x2,3x3): Y = A^T [ (G g G^T) . (B^T d B) ] A
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32, device=dev)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32, device=dev)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32, device=dev)
xpad = torch.nn.functional.pad(xd, (0, 0, 1, 1, 1, 1))                      # (N, 34, 34, C)
tiles = xpad.unfold(1, 4, 2).unfold(2, 4, 2)                               # (N, 16, 16, C, 4, 4)
V = torch.einsum('ai,nhwcij,bj->abnhwc', Bt, tiles, Bt).reshape(16, -1, C).contiguous()    # (16, T, C), T = N 16 16
U = torch.einsum('ai,kijc,bj->abkc', G, wd, G).reshape(16, K, C).contiguous()               # (16, K, C)
T = V.shape[1]
Mx = torch.empty(16, T, K, device=dev)
for xi in range(16):
    vx = V[xi].view(1, T, 1, C).contiguous()
    ux = U[xi].view(K, 1, 1, C).contiguous()
    vp, va = planes_of(vx)                       # the scale of THIS xi's transformed input (taken after the transform)
    ua = fn.absmax(ux); up = fn.weight_planes(ux, math="f16x2", w_absmax=ua)
    Mx[xi] = fn.conv2d_forward(vp, ux, None, 1, 0, 1, w_planes=up, x_absmax=va, w_absmax=ua, x_planes=True).view(T, K)
Yt = torch.einsum('ai,ijnhwk,bj->nhawbk', At, Mx.view(4, 4, N, 16, 16, K), At).reshape(N, H, W, K)
err_wino = float((Yt.double().cpu() - ref).abs().max()) / scale
# the same with the xi-GEMMs in float64 (the error the fp32 transforms alone leave)
M64 = torch.einsum('xtc,xkc->xtk', V.double().cpu(), U.double().cpu())
Y64 = torch.einsum('ai,ijnhwk,bj->nhawbk', At.double().cpu(), M64.view(4, 4, N, 16, 16, K), At.double().cpu()).reshape(N, H, W, K)
err_wino_tr = float((Y64 - ref).abs().max()) / scale

# ---- the GEMM part as one launch over all 16 x T rows (the bytes and multiply-adds of a grouped launch of the 16 GEMMs)
va_all = V.view(1, 16 * T, 1, C)
vp_all, va_ = planes_of(va_all)
u0 = U[5].view(K, 1, 1, C).contiguous(); u0a = fn.absmax(u0); u0p = fn.weight_planes(u0, math="f16x2", w_absmax=u0a)
m_all = torch.empty(1, 16 * T, 1, K, device=dev)
t_gemm_planes = timeit(lambda: fn.conv2d_forward(vp_all, u0, None, 1, 0, 1, w_planes=u0p, x_absmax=va_, w_absmax=u0a, x_planes=True, out=m_all))
vaf = fn.absmax(va_all)
t_gemm_float = timeit(lambda: fn.conv2d_forward(va_all, u0, None, 1, 0, 1, w_planes=u0p, x_absmax=vaf, w_absmax=u0a, out=m_all))
# the 16 GEMMs as 16 launches of 128 tiles each (what exists without a grouped kernel)
vps = [planes_of(V[xi].view(1, T, 1, C).contiguous()) for xi in range(16)]
mo = torch.empty(1, T, 1, K, device=dev)
def sixteen():
    for xi in range(16):
        fn.conv2d_forward(vps[xi][0], u0, None, 1, 0, 1, w_planes=u0p, x_absmax=vps[xi][1], w_absmax=u0a, x_planes=True, out=mo)
t_sixteen = timeit(sixteen)
in_b, out_b = N * H * W * C * 4, N * H * W * K * 4
print("layer: %d x %d x %d x %d -> %d, 3x3/1: direct 19.3 GMAC, Winograd 8.6 GMAC in 16 GEMMs of %d x %d x %d" % (N, H, W, C, K, T, C, K))
print("direct (conv_ntw_kernel, x as piece planes)                 %7.1f us   max error / max |y| vs float64: %.2e" % (t_direct, err_direct))
print("fp32 MFMA (the yardstick of the error)                                     max error: %.2e" % err_fp32)
print("Winograd, two-piece xi-GEMMs scaled after the transform, fp32 transforms   max error: %.2e  (%.1f x direct, %.1f x fp32 MFMA; fp32 transforms alone with exact GEMMs: %.2e)"
      % (err_wino, err_wino / err_direct, err_wino / err_fp32, err_wino_tr))
print("xi-GEMMs as ONE plane-fed launch over 16 x %d rows          %7.1f us   = %.2f of direct  (+ unfused transforms: %.0f MB written + %.0f MB read again at 5.6 TB/s = %.0f us)"
      % (T, t_gemm_planes, t_gemm_planes / t_direct, 4 * in_b / 1e6 + 4 * out_b / 1e6, 4 * in_b / 1e6 + 4 * out_b / 1e6, (in_b + 4 * in_b + 4 * in_b + 4 * out_b + 4 * out_b + out_b) / 5.6e6))
print("the same GEMM with the A operand cut in the loader (fused form) %7.1f us   = %.2f of direct  (without the transform's 2 x 32 additions per 16 values and the output transform)" % (t_gemm_float, t_gemm_float / t_direct))
print("the 16 GEMMs as 16 launches of 128 tiles                       %7.1f us   = %.2f of direct" % (t_sixteen, t_sixteen / t_direct))
