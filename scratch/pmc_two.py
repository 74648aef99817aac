import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
for (N, H, W, Cin, Cout, k) in [(32, 32, 32, 256, 256, 3), (32, 32, 32, 1024, 1024, 1), (32, 64, 64, 128, 128, 3), (32, 32, 32, 2304, 256, 1)]:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    for _ in range(3):
        fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out)
    torch.cuda.synchronize()
