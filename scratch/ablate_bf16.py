"""timing-only ablations of conv_nt_kernel on bf16 TENSORS (the *_bf16 build); needs the ABLATE build:
make -C dspnet_amd/csrc ABLATE=1; DSPN_LIB=dspnet_amd/libdspn_hip_ablate.so python scratch/ablate_bf16.py"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
shapes = [(32, 32, 32, 256, 256, 3), (32, 64, 64, 128, 128, 3), (32, 32, 32, 1024, 256, 1), (32, 128, 128, 256, 128, 1)]
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"
for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16(); w = (torch.randn(Cout, k, k, Cin, device=dev) * 0.05).bfloat16()
    out = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    kw = {}
    if fused:
        sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
        tiles, _ = fn.conv_stats_layout(N * H * W, Cout)
        kw.update(in_affine=(sc, sh, True), out_stats=torch.empty(tiles, 2, Cout, device=dev))
    res = []
    for bits, nm in ((0, "full"), (32, "no-epi"), (32 + 128, "no-epi,no-loads"), (32 + 2, "no-epi,no-lds-store"),
                     (32 + 2 + 128, "no-epi,no-loads,no-store"), (32 + 2 + 4 + 128, "..+no-barrier")):
        fn.L().dspn_debug_set_bf16(bits)
        t = timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out, **kw))
        res.append("%s %.3f" % (nm, t))
    fn.L().dspn_debug_set_bf16(0)
    bytes_ = (x.numel() + out.numel() + w.numel()) * 2
    print((N, H, W, Cin, Cout, k), "fused" if fused else "plain", "MFMA floor at 2.5PF %.3f ms, HBM floor at 5 TB/s %.3f ms |" % (fl / 2.5e12, bytes_ / 5e9), " | ".join(res), flush=True)
