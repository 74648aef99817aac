"""two-piece fp16 math against the three-piece bf16 split and the fp32 MFMA, all against float64: forward / data gradient of
a few layers (unit scales: operands N(0, 1)-sized)"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, torch.nn.functional as F
from dspnet_amd import functional as fn
for (N, H, Cin, Cout, k) in [(8, 32, 256, 256, 3), (8, 64, 128, 512, 1), (8, 64, 64, 64, 3), (2, 32, 36, 40, 3)]:
    g = torch.Generator().manual_seed(N + H + Cin)
    x = torch.randn(N, Cin, H, H, generator=g, dtype=torch.float64).float().double().requires_grad_()
    w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)).float().double()
    y_ref = F.conv2d(x, w, None, 1, k // 2)
    dy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64).float().double()
    y_ref.backward(dy)
    cp = fn.pad4(Cin)
    xd = torch.zeros(N, H, H, cp); xd[..., :Cin] = x.detach().permute(0, 2, 3, 1).float(); xd = xd.cuda()
    wd = torch.zeros(Cout, k, k, cp); wd[..., :Cin] = w.permute(0, 2, 3, 1).float(); wd = wd.cuda()
    dyd = dy.permute(0, 2, 3, 1).float().contiguous().cuda()
    if dyd.shape[3] % 4: dyd = F.pad(dyd, (0, 4 - dyd.shape[3] % 4))
    def rel(got, exp):
        d = got.double().cpu() - exp
        return float(d.abs().max() / exp.abs().max()), float((d * d).mean().sqrt() / exp.abs().max())
    out = {}
    for mode in ("fp32", "bf16x3", "f16x2"):
        fn.set_conv_math(mode)
        y = fn.conv2d_forward(xd, wd, None, 1, k // 2, 1)
        dx = fn.conv2d_dgrad(dyd, fn.weight_transpose(wd), tuple(xd.shape), 1, k // 2, 1)
        out[mode] = (rel(y.permute(0, 3, 1, 2)[:, :Cout], y_ref.detach()), rel(dx.permute(0, 3, 1, 2)[:, :Cin], x.grad))
    print((N, H, Cin, Cout, k), {m: "fwd max %.2e rms %.2e | dgrad max %.2e rms %.2e" % (v[0] + v[1]) for m, v in out.items()})
