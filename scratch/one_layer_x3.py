"""one split-math layer, N launches, for counter passes: python scratch/one_layer_x3.py H Cin Cout k [fwd|fused|dgrad|wgrad] [reps]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
H, Cin, Cout, k = [int(v) for v in sys.argv[1:5]]
what = sys.argv[5] if len(sys.argv) > 5 else "fwd"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
B = 32
fn.set_conv_math("bf16x3")
x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
dy = torch.randn(B, H, H, Cout, device="cuda"); y = torch.empty(B, H, H, Cout, device="cuda"); dx = torch.empty_like(x)
dw = torch.empty_like(w)
wt = fn.weight_transpose(w); wp = fn.weight_planes(w, math="bf16x3"); wtp = fn.weight_planes(w, transposed=True, cols=Cout, math="bf16x3")
sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda")
tiles, _ = fn.conv_stats_layout(B * H * H, Cout)
st = torch.empty(tiles, 2, Cout, device="cuda")
for _ in range(reps):
    if what == "fwd": fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=y, w_planes=wp)
    elif what == "fused": fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=y, w_planes=wp, in_affine=(sc, sh, True), out_stats=st)
    elif what == "dgrad": fn.conv2d_dgrad(dy, wt, tuple(x.shape), 1, k // 2, 1, out=dx, wt_planes=wtp)
    else: fn.conv2d_wgrad(x, dy, tuple(w.shape), 1, k // 2, 1, out=dw)
torch.cuda.synchronize()
