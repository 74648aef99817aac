"""per-conv-layer timing on the GPU: fwd / dgrad / wgrad TFLOP/s (algorithmic)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import engine as E, functional as fn, synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
import os
if os.environ.get("DSPN_WIDE_TILES"):      # force one tile shape of the wide family (dspn_conv_set_wide_tiles)
    from dspnet_amd import _lib
    _lib.lib().dspn_conv_set_wide_tiles(int(os.environ["DSPN_WIDE_TILES"]))
net = get_multi_symbol_train("resnet-50", 512, num_classes=8, batch_size=B, device=dev)
gen = synthetic.rng(1)
net.data.data.copy_(torch.from_numpy(synthetic.images(B, 512, 512, gen)))
net.label_det.data.copy_(torch.from_numpy(synthetic.det_labels(B, gen=gen)))
net.label_seg.data.copy_(torch.from_numpy(synthetic.seg_labels(B, gen=gen)))
net.g.forward(); net.g.begin_backward(); torch.cuda.synchronize()
PEAK = {"f16x2": 833.3, "bf16x3": 416.7, "fp32": 157.3, "bf16": 2500.0}[net.g.math]
print("math", net.g.math, "peak", PEAK)
def timeit(f, reps=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
# round 5: operands as the GRAPH hands them over -- dy as fp16 piece planes where the BatchNorm behind the convolution writes its
# dx that way (BatchNorm.dx_planes), x as planes where Graph._plan_input_planes materialised them -- and, per layer, the
# algorithmic HBM bytes of each pass (x or dy read once, y or dx written once, weights) with the bandwidth they imply
planes_dy = {id(b.x): True for b in net.g.nodes if isinstance(b, E.BatchNorm) and getattr(b, "dx_planes", False)}
def as_planes(t):
    am = fn.absmax(t)
    return fn.bn_apply_planes(t, torch.ones(t.shape[-1], device=dev), torch.zeros(t.shape[-1], device=dev), am), am
tot = [0, 0, 0]; totf = [0, 0, 0]
seen = set(); cnt = {}; rows = {}
print("%-36s %9s %5s %6s | %8s %6s %4s %4s | %8s %6s %4s %4s | %8s %6s %4s %4s" % ("layer", "M", "N", "K", "fwd ms", "TF", "frac", "TB/s", "dgrad ms", "TF", "frac", "TB/s", "wgrad ms", "TF", "frac", "TB/s"))
for n in net.g.nodes:
    if not isinstance(n, E.Conv): continue
    N_, H, W, Cin = n.x.shape
    Cout, R, S, _ = n.w.shape
    M = n.out.shape[0] * n.out.shape[1] * n.out.shape[2]
    key = (M, Cout, R, Cin, n.stride)
    fl = n.flops_fwd
    tf = timeit(n.forward)
    dy = n.out.own_grad()
    # "f16x2": the operand magnitudes are taken outside the timed calls, as the graph does (fused into the producers)
    f16 = n.math == "f16x2"
    xa, wa = (n._magnitudes("x"), n._magnitudes("w")) if f16 else (None, None)
    dya = fn.absmax(dy) if f16 else None
    dyp = False
    if f16 and planes_dy.get(id(n.out)) and dy.dtype == torch.float32 and dy.shape[-1] % 32 == 0:
        dy.normal_(); dy, dya = as_planes(dy); dyp = True
    xp = n._x_planes() if (f16 and hasattr(n, "_x_planes")) else None
    if xp is not None:
        tw = timeit(lambda: fn.conv2d_wgrad(xp, dy, n.w.shape, n.stride, n.pad, n.dil, out=n.w.grad, math=n.math, x_absmax=xa, dy_absmax=dya, dy_planes=dyp, x_planes=True))
    else:
        tw = timeit(lambda: fn.conv2d_wgrad(n.x_raw.data, dy, n.w.shape, n.stride, n.pad, n.dil, out=n.w.grad, in_affine=n.in_affine, math=n.math, x_absmax=xa, dy_absmax=dya, dy_planes=dyp))
    td = None
    fused_d = 0
    if n.x.requires_grad:
        dx = n.x.own_grad()
        # round 6: the data gradient as the GRAPH calls it -- with the BatchNorm-backward sums of the layer in front in its
        # epilogue (that epilogue also reads the BatchNorm's input rows: counted in the bytes below)
        bn = getattr(n, "bn_bwd_node", None)
        bn_bwd = bn_dya = None
        if bn is not None and bn.bwd_sums is not None and n.math == "f16x2":
            bn_bwd = (bn.x.data, bn.scale, bn.shift, bn.mean, bn.rstd, bn.relu, bn.bwd_sums[0])
            bn_dya = net.g.scalar(bn.am_dyin) if getattr(bn, "dx_planes", False) else None
            fused_d = 1
        td = timeit(lambda: fn.conv2d_dgrad(dy, n.wt, n.x.shape, n.stride, n.pad, n.dil, out=dx, wt_planes=n.wtp, math=n.math, dy_absmax=dya, w_absmax=wa, wt_shape=n.wt_shape, dy_planes=dyp, bn_bwd=bn_bwd, bn_dy_absmax=bn_dya))
    tot[0] += tf; tot[2] += tw; totf[0] += fl; totf[2] += fl
    if td: tot[1] += td; totf[1] += fl
    xin = N_ * H * W * Cin * 4 // (n.stride * n.stride if R * S == 1 else 1)          # (a strided 1x1 reads every stride-th pixel)
    byt = xin + M * Cout * 4 + Cout * R * S * Cin * 4
    # the bytes the FUSED calls move on top: forward + residual (the shortcut rows the epilogue adds), data gradient + the rows
    # of the BatchNorm input its epilogue reads for the backward sums
    byt_f = byt + (M * Cout * 4 if n.residual is not None else 0)
    byt_d = byt + (xin if fused_d else 0)
    key = key + (n.residual is not None, fused_d)
    cnt[key] = cnt.get(key, 0) + 1
    rows.setdefault(key, (n.w.name[:-7] + ("*" if dyp else "") + ("+" if xp is not None else "") + ("r" if n.residual is not None else "") + ("^" if fused_d else ""),
                          M, Cout, R * S * Cin, tf, fl, td, tw, byt, byt_f, byt_d))
print("(* dy as piece planes in dgrad / wgrad, + x as piece planes in forward / wgrad, r forward adds a residual in its epilogue, ^ data gradient with the")
print(" BatchNorm-backward sums in its epilogue; TB/s = bytes the call moves / time, of 8: operand + output + weights, + the residual rows (r), + the BatchNorm input rows (^))")
for key, (name, M, Cout, K, tf, fl, td, tw, byt, byt_f, byt_d) in rows.items():
    print("%-32s x%-2d %8d %5d %6d | %7.3f %6.1f %4.2f %4.1f | %7s %6s %4s %4s | %7.3f %6.1f %4.2f %4.1f" % (
        name, cnt[key], M, Cout, K, tf, fl / tf / 1e9, fl / tf / 1e9 / PEAK, byt_f / tf / 1e9,
        ("%.3f" % td) if td else "-", ("%.1f" % (fl / td / 1e9)) if td else "-", ("%.2f" % (fl / td / 1e9 / PEAK)) if td else "-",
        ("%.1f" % (byt_d / td / 1e9)) if td else "-", tw, fl / tw / 1e9, fl / tw / 1e9 / PEAK, byt / tw / 1e9))
print("TOTAL fwd %.2f ms (%.1f TF)  dgrad %.2f ms (%.1f TF)  wgrad %.2f ms (%.1f TF)" % (
    tot[0], totf[0] / tot[0] / 1e9, tot[1], totf[1] / tot[1] / 1e9, tot[2], totf[2] / tot[2] / 1e9))
