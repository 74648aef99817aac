import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn, _lib
L = _lib.lib()
for hw, theta in (((4, 4), (0.98, 0.03, -0.02, -0.04, 1.05, 0.01)), ((16, 16), (0.98, 0.03, -0.02, -0.04, 1.05, 0.01)), ((64, 64), (0.97, 0.04, -0.03, -0.05, 1.04, 0.02)), ((16, 16), (1, 0, 0, 0, 1, 0))):
    g = torch.Generator().manual_seed(3)
    B, C, Ho, Wo = 2, 8, 16 if hw[0] <= 16 else 64, 12 if hw[0] <= 16 else 64
    th = torch.tensor(theta, dtype=torch.float32, device="cuda")
    x = torch.randn(B, hw[0], hw[1], C, generator=g).cuda()
    dy = torch.randn(B, Ho, Wo, C + 4, generator=g).cuda()
    rows = fn.affine_sampler_theta_rows(x.shape, Ho)
    out = {}
    for on in (0, 1):
        L.dspn_affine_sampler_set_batched(on)
        part = torch.zeros((rows, 6), dtype=torch.float64, device="cuda")
        dx = fn.affine_sampler_backward_data_theta(dy, th, x, 0, part)
        out[on] = (dx, part)
    d = (out[0][0] - out[1][0]).abs()
    print(hw, theta[:2], "dx max diff %.3e of %.3e; mismatching elements %d of %d; theta rows max diff %.3e" % (
        float(d.max()), float(out[0][0].abs().max()), int((d > 0).sum()), d.numel(), float((out[0][1] - out[1][1]).abs().max())))
    bad = (d.amax(dim=(0, 3)) > 0).nonzero()
    print("   source positions with a difference:", bad[:12].cpu().tolist())
