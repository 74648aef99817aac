"""timing-only ablations of conv_nt_kernel in the split (bf16x3) mode; needs the ABLATE build:
make -C dspnet_amd/csrc ABLATE=1; DSPN_LIB=dspnet_amd/libdspn_hip_ablate.so python scratch/ablate_split.py"""
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
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
fn.set_conv_math(mode)
fused = len(sys.argv) > 2 and sys.argv[2] == "fused"
for (N, H, W, Cin, Cout, k) in shapes:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    kw = {}
    if mode == "f16x2":       # magnitudes and planes as the graph passes them (made once, outside the timed calls)
        kw = dict(x_absmax=fn.absmax(x), w_absmax=fn.absmax(w))
    if mode in ("f16x2", "bf16x3") and Cin % 32 == 0:
        kw["w_planes"] = fn.weight_planes(w, math=mode, w_absmax=kw.get("w_absmax"))
    if fused:
        sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
        tiles, _ = fn.conv_stats_layout(N * H * W, Cout)
        kw.update(in_affine=(sc, sh, True), out_stats=torch.empty(tiles, 2, Cout, device=dev))
        if mode == "f16x2":
            kw["x_absmax"] = fn.absmax(x, (sc, sh, True)); kw["out_minmax"] = torch.empty(tiles, 2, Cout, device=dev)
    nprod = {"f16x2": 3, "bf16x3": 6}.get(mode, 1)
    res = []
    for bits, nm in ((0, "full"), (32, "no-epi"), (32 + 64, "no-epi,1 product"), (32 + 128, "no-epi,no-loads"), (32 + 2, "no-epi,no-lds-store"),
                     (32 + 2 + 128, "no-epi,no-loads,no-store"), (32 + 2 + 4 + 128, "..+no-barrier"), (32 + 2 + 4 + 128 + 64, "..+1 product"), (256, "no piece arithmetic"), (32 + 256, "no-epi,no piece arithmetic"), (4096, "A side every 3rd k-step"), (8192, "A side every 9th k-step"), (32 + 4096, "no-epi, A every 3rd"), (32 + 8192, "no-epi, A every 9th")):
        fn.L().dspn_debug_set(bits)
        t = timeit(lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out, **kw))
        res.append("%s %.3f" % (nm, t))
    fn.L().dspn_debug_set(0)
    print((N, H, W, Cin, Cout, k), "fused" if fused else "plain", "%d-product MFMA floor at 2.5PF %.3f ms |" % (nprod, nprod * fl / 2.5e12), " | ".join(res), flush=True)
