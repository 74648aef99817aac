"""bf16-tensor convolutions per resnet-50 layer: measured time next to the two floors of the launch --
algorithmic HBM bytes / 8 TB/s and flops / 2.5 PFLOP/s -- for forward, data gradient, weight gradient"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
from conv_modes import LAYERS, timeit
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = torch.bfloat16
print("%-12s %8s | %-30s | %-30s | %-30s" % ("layer", "GFLOP", "fwd ms (hbm-floor mfma-floor x)", "dgrad", "wgrad"))
tot = [0, 0, 0, 0, 0, 0]
for name, H, W, Cin, Cout, k, stride, pad in LAYERS:
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    fl = 2.0 * B * Ho * Wo * Cin * Cout * k * k
    cp, kp = fn.padc(Cin, dt), fn.padc(Cout, dt)
    x = torch.randn(B, H, W, cp, device="cuda").to(dt)
    w32 = torch.randn(Cout, k, k, cp, device="cuda") * 0.05
    dy = torch.randn(B, Ho, Wo, kp, device="cuda").to(dt)
    y = torch.empty(B, Ho, Wo, kp, device="cuda", dtype=dt)
    dx = torch.empty_like(x); dw = torch.empty_like(w32)
    wt = fn.weight_transpose(w32, dtype=dt); wop = w32.to(dt)
    tf = timeit(lambda: fn.conv2d_forward(x, wop, None, stride, pad, 1, out=y), 10)
    td = timeit(lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=dx), 10) if Cin > 8 else float("nan")
    tw = timeit(lambda: fn.conv2d_wgrad(x, dy, tuple(w32.shape), stride, pad, 1, out=dw), 10)
    bx, by, bw = x.numel() * 2, y.numel() * 2, w32.numel() * 2
    hb = (bx + by + bw) / 8e12 * 1e3
    hbw = (bx + by + w32.numel() * 4) / 8e12 * 1e3
    mf = fl / 2.5e15 * 1e3
    fl_floor, w_floor = max(hb, mf), max(hbw, mf)
    def cell(t, floor, h):
        return "%6.3f (%5.3f %5.3f) x%4.1f" % (t, h, mf, t / floor)
    print("%-12s %8.1f | %-30s | %-30s | %-30s" % (name, fl / 1e9, cell(tf, fl_floor, hb), cell(td, fl_floor, hb), cell(tw, w_floor, hbw)))
    tot[0] += tf; tot[1] += 0 if td != td else td; tot[2] += tw; tot[3] += fl_floor; tot[4] += w_floor
    del x, dy, y, dx
print("sum of the listed layers: fwd %.2f dgrad %.2f wgrad %.2f ms; floors fwd/dgrad %.2f wgrad %.2f" % tuple(tot[:5]))
