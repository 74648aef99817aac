"""per-conv-layer timing on the GPU: fwd / dgrad / wgrad TFLOP/s (algorithmic)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import engine as E, functional as fn, synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
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
tot = [0, 0, 0]; totf = [0, 0, 0]
seen = set(); cnt = {}; rows = {}
print("%-34s %9s %5s %6s | %8s %6s | %8s %6s | %8s %6s" % ("layer", "M", "N", "K", "fwd ms", "TF", "dgrad ms", "TF", "wgrad ms", "TF"))
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
    tw = timeit(lambda: fn.conv2d_wgrad(n.x_raw.data, dy, n.w.shape, n.stride, n.pad, n.dil, out=n.w.grad, in_affine=n.in_affine, math=n.math, x_absmax=xa, dy_absmax=dya))
    td = None
    if n.x.requires_grad:
        dx = n.x.own_grad()
        td = timeit(lambda: fn.conv2d_dgrad(dy, n.wt, n.x.shape, n.stride, n.pad, n.dil, out=dx, wt_planes=n.wtp, math=n.math, dy_absmax=dya, w_absmax=wa, wt_shape=n.wt_shape))
    tot[0] += tf; tot[2] += tw; totf[0] += fl; totf[2] += fl
    if td: tot[1] += td; totf[1] += fl
    cnt[key] = cnt.get(key, 0) + 1
    rows.setdefault(key, (n.w.name[:-7], M, Cout, R * S * Cin, tf, fl, td, tw))
for key, (name, M, Cout, K, tf, fl, td, tw) in rows.items():
    print("%-30s x%-2d %8d %5d %6d | %7.3f %6.1f %4.2f | %7s %6s %4s | %7.3f %6.1f %4.2f" % (
        name, cnt[key], M, Cout, K, tf, fl / tf / 1e9, fl / tf / 1e9 / PEAK,
        ("%.3f" % td) if td else "-", ("%.1f" % (fl / td / 1e9)) if td else "-", ("%.2f" % (fl / td / 1e9 / PEAK)) if td else "-",
        tw, fl / tw / 1e9, fl / tw / 1e9 / PEAK))
print("TOTAL fwd %.2f ms (%.1f TF)  dgrad %.2f ms (%.1f TF)  wgrad %.2f ms (%.1f TF)" % (
    tot[0], totf[0] / tot[0] / 1e9, tot[1], totf[1] / tot[1] / 1e9, tot[2], totf[2] / tot[2] / 1e9))
