"""one 3x3 layer (32 x 32 x 32 x 256 -> 256) through forward (plain / fused), data gradient and weight gradient, for SQ counter passes"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
N, H, W, Cin, Cout, k = 32, 32, 32, 256, 256, 3
x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
dy = torch.randn(N, H, W, Cout, device=dev)
o = torch.empty(N, H, W, Cout, device=dev)
aff = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev), True)
tiles, _ = fn.conv_stats_layout(N * H * W, Cout)
st = torch.empty(tiles, 2, Cout, device=dev)
wt = fn.weight_transpose(w)
nsp = fn.L().dspn_conv2d_wgrad_splits(N, H, W, Cin, Cout, k, k, 1)
slabs = torch.empty(nsp, Cout, k, k, Cin, device=dev)
for _ in range(3):
    fn.conv2d_forward(x, w, None, 1, 1, 1, out=o)
    fn.conv2d_forward(x, w, None, 1, 1, 1, out=o, in_affine=aff, out_stats=st)
    fn.conv2d_dgrad(dy, wt, x.shape, 1, 1, 1, out=o)
    fn.conv2d_wgrad_slabs(x, dy, (Cout, k, k, Cin), slabs, 1, 1, 1)
    fn.conv2d_wgrad_slabs(x, dy, (Cout, k, k, Cin), slabs, 1, 1, 1, in_affine=aff)
torch.cuda.synchronize()
