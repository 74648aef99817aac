"""where the cycles of a k-step go inside conv_nt_kernel (two-piece math): per-wave s_memtime stamps summed per phase
(ABLATE build, dspn_debug_set bit 16384):  DSPN_LIB=dspnet_amd/libdspn_hip_ablate.so python scratch/phase_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dspnet_amd import functional as fn
fn.set_conv_math("f16x2")
dev = torch.device("cuda", 0)
L = fn.L()
L.dspn_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
NAMES = ["issue loads", "MFMA block 0", "wait rows", "pieces", "MFMA block 1", "LDS stores", "barrier"]
def run(name, f):
    for _ in range(200): f()
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, np.uint32)
    L.dspn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
    L.dspn_debug_set(16384); f(); torch.cuda.synchronize(); L.dspn_debug_set(0)
    L.dspn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
    v = buf.reshape(-1, 8).astype(np.float64)
    v = v[v[:, 7] > 0]
    per = v[:, :7] / v[:, 7:8]
    tot = per.sum(1)
    print("%s: %d waves, %.0f k-steps each, %.0f cycles per k-step (median; 5%%..95%% %.0f..%.0f)" % (
        name, len(v), np.median(v[:, 7]), np.median(tot), np.percentile(tot, 5), np.percentile(tot, 95)))
    for half, sel in (("waves 0-3", np.arange(len(v)) % 8 < 4), ("waves 4-7", np.arange(len(v)) % 8 >= 4)):
        print("   %s: " % half + "  ".join("%s %.0f" % (n, np.median(per[sel, i])) for i, n in enumerate(NAMES)))
for (N, H, W, Cin, Cout, k) in [(32, 32, 32, 256, 256, 3), (32, 16, 16, 512, 512, 3), (32, 64, 64, 128, 128, 3), (32, 32, 32, 1024, 256, 1)]:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    wa = fn.absmax(w); xa = fn.absmax(x)
    wp = fn.weight_planes(w, math="f16x2", w_absmax=wa); wtp = fn.weight_planes(w, transposed=True, cols=Cout, math="f16x2", w_absmax=wa)
    wt = fn.weight_transpose(w)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
    xa2 = fn.absmax(x, (sc, sh, True))
    tiles, _ = fn.conv_stats_layout(N * H * W, Cout)
    st = torch.empty(tiles, 2, Cout, device=dev)
    dy = torch.randn(N, H, W, Cout, device=dev); dx = torch.empty_like(x); dya = fn.absmax(dy)
    run("fwd plain %s" % ((N, H, W, Cin, Cout, k),), lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out, w_planes=wp, x_absmax=xa, w_absmax=wa))
    run("fwd affine+stats", lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out, w_planes=wp, in_affine=(sc, sh, True), out_stats=st, x_absmax=xa2, w_absmax=wa))
    run("dgrad", lambda: fn.conv2d_dgrad(dy, wt, tuple(x.shape), 1, k // 2, 1, out=dx, wt_planes=wtp, dy_absmax=dya, w_absmax=wa))
