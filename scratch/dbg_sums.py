import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from dspnet_amd import functional as fn
N, H, W, Cin, Cout, k, stride, pad = (8, 128, 128, 64, 16, 1, 1, 0)
g = torch.Generator().manual_seed(1)
x = torch.randn(N, H, W, Cin, generator=g).cuda(); dy = torch.randn(N, H, W, Cout, generator=g).cuda()
w = (torch.randn(Cout, k, k, Cin, generator=g) / np.sqrt(Cin)).cuda()
gamma = (torch.rand(Cin, generator=g) + 0.5).cuda(); beta = torch.randn(Cin, generator=g).cuda()
mean, rstd, scale, shift = fn.bn_stats(x, 2e-5, gamma, beta)
wt = fn.weight_transpose(w)
d_ref = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1)
dx_ref, dg_ref, db_ref = fn.bn_backward(x, scale, shift, d_ref, mean, rstd, gamma, relu=True)
tiles = fn.conv_dgrad_bn_tiles(tuple(x.shape), stride)
sums = torch.full((tiles, 2, Cin), float("nan"), device="cuda")
d = fn.conv2d_dgrad(dy, wt, tuple(x.shape), stride, pad, 1, bn_bwd=(x, scale, shift, mean, rstd, True, sums))
dx, dg, db = fn.bn_backward_from_sums(x, scale, shift, d, mean, rstd, gamma, sums, tiles, relu=True)
# exact in float64 from d_ref
mask = (x.double() * scale.double() + shift.double()) > 0
gd = torch.where(mask, d_ref.double(), torch.zeros_like(d_ref.double()))
db64 = gd.sum(dim=(0, 1, 2)); dg64 = (gd * (x.double() - mean.double()) * rstd.double()).sum(dim=(0, 1, 2))
print("tiles", tiles)
print("db  ref err", float((db_ref.double() - db64).abs().max()), "new err", float((db.double() - db64).abs().max()), "from sums", float((sums[:, 0].double().sum(0) - db64).abs().max()))
print("dg  ref err", float((dg_ref.double() - dg64).abs().max()), "new err", float((dg.double() - dg64).abs().max()), "from sums", float((sums[:, 1].double().sum(0) - dg64).abs().max()))
print("dx err", float((dx - dx_ref).abs().max()))
