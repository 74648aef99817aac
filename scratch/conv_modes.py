"""resnet-50 layer shapes (B=32, 512x512): forward / dgrad / wgrad time in the three modes of the convolution family --
fp32 tensors + fp32 MFMA, fp32 tensors + bf16 MFMA (rounded on the way into LDS), bf16 tensors + bf16 MFMA"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LAYERS = [  # name, H, W, Cin, Cout, k, stride, pad
    ("conv0", 512, 512, 3, 64, 7, 2, 3),
    ("s1_conv1", 128, 128, 64, 64, 1, 1, 0), ("s1_conv2", 128, 128, 64, 64, 3, 1, 1), ("s1_conv3", 128, 128, 64, 256, 1, 1, 0),
    ("s1_u2conv1", 128, 128, 256, 64, 1, 1, 0),
    ("s2_conv1", 128, 128, 256, 128, 1, 1, 0), ("s2_conv2s2", 128, 128, 128, 128, 3, 2, 1), ("s2_conv2", 64, 64, 128, 128, 3, 1, 1),
    ("s2_conv3", 64, 64, 128, 512, 1, 1, 0), ("s2_u2conv1", 64, 64, 512, 128, 1, 1, 0),
    ("s3_conv2", 32, 32, 256, 256, 3, 1, 1), ("s3_conv3", 32, 32, 256, 1024, 1, 1, 0), ("s3_u2conv1", 32, 32, 1024, 256, 1, 1, 0),
    ("s4_conv2", 16, 16, 512, 512, 3, 1, 1), ("s4_conv3", 16, 16, 512, 2048, 1, 1, 0), ("s4_u2conv1", 16, 16, 2048, 512, 1, 1, 0),
]
def timeit(f, reps=5):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("%-12s %10s | %s" % ("layer", "GFLOP", "  ".join("%-26s" % m for m in ("fp32: fwd dgrad wgrad ms", "bf16 math: fwd dgrad wgrad", "bf16 tensors: fwd dgrad wgrad"))))
tot = {}
for name, H, W, Cin, Cout, k, stride, pad in LAYERS:
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    fl = 2.0 * B * Ho * Wo * Cin * Cout * k * k
    row = []
    for mode in ("fp32", "bf16math", "bf16"):
        dt = torch.bfloat16 if mode == "bf16" else torch.float32
        cp, kp = fn.padc(Cin, dt), fn.padc(Cout, dt)
        x = torch.randn(B, H, W, cp, device="cuda").to(dt)
        w32 = torch.randn(Cout, k, k, cp, device="cuda") * 0.05
        dy = torch.randn(B, Ho, Wo, kp, device="cuda").to(dt)
        y = torch.empty(B, Ho, Wo, kp, device="cuda", dtype=dt)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w32)
        wt = fn.weight_transpose(w32, dtype=dt)
        wop = w32.to(dt)
        fn.set_conv_math("bf16" if mode == "bf16math" else "fp32")
        tf = timeit(lambda: fn.conv2d_forward(x, wop, None, stride, pad, 1, out=y))
        td = timeit(lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=dx)) if Cin > 8 else float("nan")
        tw = timeit(lambda: fn.conv2d_wgrad(x, dy, tuple(w32.shape), stride, pad, 1, out=dw))
        fn.set_conv_math("fp32")
        row.append((tf, td, tw))
        t = tot.setdefault(mode, [0.0, 0.0, 0.0, 0.0])
        t[0] += tf; t[1] += 0 if td != td else td; t[2] += tw; t[3] += fl
        del x, dy, y, dx
    print("%-12s %10.1f | %s" % (name, fl / 1e9, "  ".join("%7.3f %7.3f %7.3f    " % r for r in row)))
for mode, t in tot.items():
    print("%-9s total fwd %.2f ms (%.0f TF)  dgrad %.2f  wgrad %.2f ms (%.0f TF)" % (mode, t[0], t[3] / t[0] / 1e9, t[1], t[2], t[3] / t[2] / 1e9))
