"""a few representative conv launches (fwd, dgrad, wgrad) for PMC collection"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
shapes = [(32, 32, 32, 256, 256, 3), (32, 64, 64, 128, 128, 3), (32, 128, 128, 64, 256, 1), (32, 64, 64, 128, 512, 1), (32, 32, 32, 1024, 256, 1), (32, 128, 128, 64, 64, 3)]
for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    dy = torch.randn(N, H, W, Cout, device=dev)
    for _ in range(2):
        fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out)
        wt = fn.weight_transpose(w)
        fn.conv2d_dgrad(dy, wt, tuple(x.shape), 1, k // 2, 1)
        fn.conv2d_wgrad(x, dy, tuple(w.shape), 1, k // 2, 1)
    torch.cuda.synchronize()
