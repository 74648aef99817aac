"""resnet-50 layer shapes (B=32, 512x512) in the split math (DSPN_MATH_F32_BF16X3): forward (plain, and with the fused
input affine + output statistics), data gradient, weight gradient -- ms, TFLOP/s and fraction of 416.7 (= 2500 / 6).
Weight piece planes are made once outside the timed region (the graph makes them once per step for all layers).
DSPN_NT_NOHALO=1 python scratch/x3_layers.py = the same with the generic 3x3 kernel (A/B on one box)."""
import os, sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
only = sys.argv[2] if len(sys.argv) > 2 else None
LAYERS = [  # name, H, W, Cin, Cout, k, stride, pad
    ("s1_conv1", 128, 128, 64, 64, 1, 1, 0), ("s1_conv2", 128, 128, 64, 64, 3, 1, 1), ("s1_conv3", 128, 128, 64, 256, 1, 1, 0),
    ("s1_u2conv1", 128, 128, 256, 64, 1, 1, 0),
    ("s2_conv1", 128, 128, 256, 128, 1, 1, 0), ("s2_conv2s2", 128, 128, 128, 128, 3, 2, 1), ("s2_conv2", 64, 64, 128, 128, 3, 1, 1),
    ("s2_conv3", 64, 64, 128, 512, 1, 1, 0), ("s2_u2conv1", 64, 64, 512, 128, 1, 1, 0),
    ("s3_conv2s2", 64, 64, 256, 256, 3, 2, 1),
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
WG_MODE = os.environ.get("X3_WG_MODE", os.environ.get("X3_MODE", "bf16x3"))
MODE = os.environ.get("X3_MODE", "bf16x3")      # bf16x3 | f16x2
fn.set_conv_math(MODE)
print("math:", MODE)
print("%-11s %7s | %-22s | %-22s | %-22s | %-22s" % ("layer", "GFLOP", "fwd ms  TF  frac", "fwd+affine+stats", "dgrad", "wgrad"))
tot = [0.0] * 4; totf = 0.0
for name, H, W, Cin, Cout, k, stride, pad in LAYERS:
    if only and only not in name: continue
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    fl = 2.0 * B * Ho * Wo * Cin * Cout * k * k
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
    dy = torch.randn(B, Ho, Wo, Cout, device="cuda")
    y = torch.empty(B, Ho, Wo, Cout, device="cuda"); dx = torch.empty_like(x); dw = torch.empty_like(w)
    wt = fn.weight_transpose(w)
    wp, wtp = (fn.weight_planes(w, math=MODE), fn.weight_planes(w, transposed=True, cols=Cout, math=MODE)) if MODE == "bf16x3" else (None, None)
    sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda")
    tiles, _ = fn.conv_stats_layout(B * Ho * Wo, Cout)
    st = torch.empty(tiles, 2, Cout, device="cuda")
    ts = [timeit(lambda: fn.conv2d_forward(x, w, None, stride, pad, 1, out=y, w_planes=wp)),
          timeit(lambda: fn.conv2d_forward(x, w, None, stride, pad, 1, out=y, w_planes=wp, in_affine=(sc, sh, True), out_stats=st)),
          timeit(lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, out=dx, wt_planes=wtp)),
          timeit(lambda: fn.conv2d_wgrad(x, dy, tuple(w.shape), stride, pad, 1, out=dw, math=WG_MODE))]
    for i in range(4): tot[i] += ts[i]
    totf += fl
    print("%-11s %7.1f | %s" % (name, fl / 1e9, " | ".join("%7.3f %6.1f %5.2f  " % (t, fl / t / 1e9, fl / t / 1e9 / 416.7) for t in ts)))
    del x, dy, y, dx
print("%-11s %7.1f | %s" % ("TOTAL", totf / 1e9, " | ".join("%7.3f %6.1f %5.2f  " % (t, totf / t / 1e9, totf / t / 1e9 / 416.7) for t in tot)))
