import sys
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from dspnet_amd import functional as fn, _lib
L = _lib.lib()
L.dspn_conv_set_wide_tiles(4)
N, H, W, Cin, Cout = 16, 64, 64, 64, 256
g = torch.Generator().manual_seed(1)
x = torch.randn(N, H, W, Cin, generator=g).cuda()
w = (torch.randn(Cout, 1, 1, Cin, generator=g) / 8).cuda()
xa = fn.absmax(x); wa = fn.absmax(w); wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
ys = []
for on in (0, 1):
    L.dspn_conv_set_tile_spanning(on)
    ys.append(fn.conv2d_forward(x, w, None, 1, 0, 1, w_planes=wp, x_absmax=xa, w_absmax=wa))
d = (ys[0] - ys[1]).abs().view(-1, 128, Cout)       # per 128-row tile
bad = (d.amax(dim=(1, 2)) > 0).nonzero().flatten().cpu().numpy()
print("tiles", d.shape[0], "bad row tiles", len(bad), bad[:40])
if len(bad):
    t = int(bad[0])
    dd = d[t]
    rows = (dd.amax(dim=1) > 0).nonzero().flatten().cpu().numpy(); cols = (dd.amax(dim=0) > 0).nonzero().flatten().cpu().numpy()
    print("tile", t, "bad rows", rows[:64], "bad cols", len(cols), cols[:16])
