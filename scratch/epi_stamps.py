"""where the cycles of a tile's EPILOGUE go inside conv_nt_kernel (two-piece math): per-wave s_memtime stamps summed per
phase (ABLATE build, dspn_debug_set bit 32768):  DSPN_LIB=dspnet_amd/libdspn_hip_ablate.so python scratch/epi_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dspnet_amd import functional as fn
fn.set_conv_math("f16x2")
dev = torch.device("cuda", 0)
L = fn.L()
L.dspn_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
NAMES = ["stage acc->LDS", "barrier", "row loop", "stats exchange", "hand-over", "k-loop"]
def run(name, f):
    for _ in range(100): f()
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, np.uint32)
    L.dspn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
    L.dspn_debug_set(32768); f(); torch.cuda.synchronize(); L.dspn_debug_set(0)
    L.dspn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
    v = buf.reshape(-1, 8).astype(np.float64)
    v = v[v[:, 7] > 0]
    per = v[:, :6] / v[:, 7:8]
    print("%s: %d waves, %.1f tiles each; per tile (median cycles): " % (name, len(v), np.median(v[:, 7])) +
          "  ".join("%s %.0f" % (n, np.median(per[:, i])) for i, n in enumerate(NAMES)) + "  | epilogue %.0f" % np.median(per[:, :5].sum(1)))
for (N, H, W, Cin, Cout, k) in [(32, 32, 32, 256, 1024, 1), (32, 128, 128, 64, 256, 1), (32, 64, 64, 128, 512, 1), (32, 32, 32, 256, 256, 3)]:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    wa = fn.absmax(w); xa = fn.absmax(x)
    wp = fn.weight_planes(w, math="f16x2", w_absmax=wa)
    tiles, _ = fn.conv_stats_layout(N * H * W, Cout)
    st = torch.empty(tiles, 2, Cout, device=dev); mm = torch.empty(tiles, 2, Cout, device=dev)
    print((N, H, W, Cin, Cout, k))
    run("  plain      ", lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out, w_planes=wp, x_absmax=xa, w_absmax=wa))
    run("  +statistics", lambda: fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out, w_planes=wp, out_stats=st, out_minmax=mm, x_absmax=xa, w_absmax=wa))
