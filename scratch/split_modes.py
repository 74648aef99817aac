"""float-tensor convolutions in the three math modes: error against a float64 reference and time, per resnet-50 layer shape"""
import sys
sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F
from dspnet_amd import functional as fn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LAYERS = [("s1_conv1", 128, 128, 64, 64, 1, 1, 0), ("s1_conv2", 128, 128, 64, 64, 3, 1, 1), ("s1_conv3", 128, 128, 64, 256, 1, 1, 0),
          ("s2_conv1", 128, 128, 256, 128, 1, 1, 0), ("s2_conv2", 64, 64, 128, 128, 3, 1, 1), ("s2_conv3", 64, 64, 128, 512, 1, 1, 0),
          ("s3_conv2", 32, 32, 256, 256, 3, 1, 1), ("s3_u2conv1", 32, 32, 1024, 256, 1, 1, 0),
          ("s4_conv2", 16, 16, 512, 512, 3, 1, 1), ("s4_conv3", 16, 16, 512, 2048, 1, 1, 0)]
def timeit(f, reps=10):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
tot = {}
for name, H, W, Cin, Cout, k, stride, pad in LAYERS:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = torch.randn(B, Ho, Wo, Cout, device="cuda")
    y = torch.empty(B, Ho, Wo, Cout, device="cuda"); dx = torch.empty_like(x); dw = torch.empty_like(w)
    wt = fn.weight_transpose(w)
    nb = min(B, 4)   # float64 reference on a slice of the batch
    xr = x[:nb].double().permute(0, 3, 1, 2).requires_grad_(); wr = w.double().permute(0, 3, 1, 2).requires_grad_()
    yr = F.conv2d(xr, wr, None, stride, pad)
    yr.backward(dy[:nb].double().permute(0, 3, 1, 2))
    y_ref, dx_ref = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)
    fl = 2.0 * B * Ho * Wo * Cin * Cout * k * k
    cells = []
    for mode in ("fp32", "bf16x3", "bf16"):
        fn.set_conv_math(mode)
        tf = timeit(lambda: fn.conv2d_forward(x, w, None, stride, pad, 1, out=y))
        td = timeit(lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=dx))
        ey = float((y[:nb].double() - y_ref).abs().max() / y_ref.abs().max())
        ed = float((dx[:nb].double() - dx_ref).abs().max() / dx_ref.abs().max())
        fn.set_conv_math("fp32")
        cells.append("%s fwd %.3f ms %5.0f TF err %.1e | dgrad %.3f ms err %.1e" % (mode, tf, fl / tf / 1e9, ey, td, ed))
        t = tot.setdefault(mode, [0.0, 0.0]); t[0] += tf; t[1] += td
    print("%-10s %6.1f GF  " % (name, fl / 1e9) + "  ||  ".join(cells))
print({m: "fwd %.2f dgrad %.2f ms" % tuple(t) for m, t in tot.items()})
