"""tile configuration for the Cout = 64 layers of stage 1 (fused variants): forced via dspn_debug_set bits 8-10"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn, _lib
dev = torch.device("cuda", 0)
def timeit(f, reps=20):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
shapes = [(32, 16, 16, 2048, 512, 1), (32, 16, 16, 512, 2048, 1), (32, 16, 16, 512, 512, 3), (32, 32, 32, 1024, 256, 1), (32, 32, 32, 256, 1024, 1), (32, 32, 32, 256, 256, 3)] if len(sys.argv) > 1 else [(32, 128, 128, 256, 64, 1), (32, 128, 128, 64, 64, 3), (32, 128, 128, 64, 64, 1), (32, 64, 64, 512, 128, 1), (32, 64, 64, 128, 128, 3)]
for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    o = torch.empty(N, H, W, Cout, device=dev)
    aff = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev), True)
    tiles, _ = fn.conv_stats_layout(N * H * W, Cout)
    row = []
    for cfg in (None, 0, 1, 2):
        _lib.lib().dspn_debug_set(0 if cfg is None else (cfg + 1) << 8)
        try:
            tl, _ = fn.conv_stats_layout(N * H * W, Cout)
            st = torch.empty(max(tl, tiles) * 4, 2, Cout, device=dev)
            t0 = timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=o))
            t1 = timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=o, in_affine=aff, out_stats=st))
            row.append("cfg %s: plain %.3f both %.3f" % ("auto" if cfg is None else cfg, t0, t1))
        except Exception as e:
            row.append("cfg %s: %s" % (cfg, str(e)[:60]))
    _lib.lib().dspn_debug_set(0)
    print((N, H, W, Cin, Cout, k), " | ".join(row))
