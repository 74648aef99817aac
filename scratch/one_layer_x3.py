"""one split-math layer, N launches, for counter passes: python scratch/one_layer_x3.py H Cin Cout k [fwd|fused|dgrad|wgrad] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspnet_amd import functional as fn
H, Cin, Cout, k = [int(v) for v in sys.argv[1:5]]
what = sys.argv[5] if len(sys.argv) > 5 else "fwd"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
B = 32
MODE = os.environ.get("DSPN_CONV_MATH", "f16x2")      # bf16x3 | f16x2
fn.set_conv_math(MODE)
x = torch.randn(B, H, H, Cin, device="cuda"); w = torch.randn(Cout, k, k, Cin, device="cuda") * 0.05
dy = torch.randn(B, H, H, Cout, device="cuda"); y = torch.empty(B, H, H, Cout, device="cuda"); dx = torch.empty_like(x)
dw = torch.empty_like(w)
wa = fn.absmax(w) if MODE == "f16x2" else None
wt = fn.weight_transpose(w); wp = fn.weight_planes(w, math=MODE, w_absmax=wa); wtp = fn.weight_planes(w, transposed=True, cols=Cout, math=MODE, w_absmax=wa)
sc, sh = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda")
tiles, _ = fn.conv_stats_layout(B * H * H, Cout)
st = torch.empty(tiles, 2, Cout, device="cuda")
for _ in range(reps):
    if what == "fwd": fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=y, w_planes=wp, w_absmax=wa)
    elif what == "fused": fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=y, w_planes=wp, w_absmax=wa, in_affine=(sc, sh, True), out_stats=st)
    elif what == "dgrad": fn.conv2d_dgrad(dy, wt, tuple(x.shape), 1, k // 2, 1, out=dx, wt_planes=wtp, w_absmax=wa)
    else: fn.conv2d_wgrad(x, dy, tuple(w.shape), 1, k // 2, 1, out=dw)
torch.cuda.synchronize()
