"""shader clock the chip holds inside conv_nt_kernel under sustained load (dspn_debug_set bit 2048)"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
for (N, H, W, Cin, Cout, k) in [(32, 32, 32, 256, 256, 3), (32, 16, 16, 512, 512, 3), (32, 32, 32, 1024, 1024, 1), (32, 128, 128, 64, 256, 1)]:
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
    out = torch.empty(N, H, W, Cout, device=dev)
    for _ in range(300):                                   # ~0.1 s of back-to-back launches before the stamped one
        fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out)
    fn.L().dspn_debug_set(2048)
    fn.conv2d_forward(x, w, None, 1, k // 2, 1, out=out)
    fn.L().dspn_debug_set(0)
    torch.cuda.synchronize()
    G = 1024          # upper bound of the persistent grid; rows beyond the real grid hold ordinary outputs and are filtered
    raw = out.flatten()[:4 * G].view(G, 4).cpu()
    v = raw[:, :2].double()
    ticks = raw[:, 2:].contiguous().view(torch.int32).to(torch.int64) & 0xffffffff
    ok = (v[:, 1] > 100) & (v[:, 1] < 1e7) & (v[:, 0] > v[:, 1] * 10) & (v[:, 0] < v[:, 1] * 40)   # 1.0 .. 4.0 GHz
    v, ticks = v[ok], ticks[ok]
    clk = (v[:, 0] / v[:, 1] * 100e6).median().item()
    dur = (v[:, 1] / 100e6).median().item()
    t0 = ticks[:, 0].min()
    st = (ticks[:, 0] - t0).double() / 100.0; en = (ticks[:, 1] - t0).double() / 100.0       # microseconds
    print("   workgroups %d: start min/median/max %.1f/%.1f/%.1f us, end min/median/max %.1f/%.1f/%.1f us, lifetime min/median/max %.1f/%.1f/%.1f us"
          % (len(st), st.min(), st.median(), st.max(), en.min(), en.median(), en.max(), (en - st).min(), (en - st).median(), (en - st).max()))
    fl = 2.0 * N * H * W * Cin * Cout * k * k
    print((N, H, W, Cin, Cout, k), "in-kernel clock %.3f GHz; workgroup lifetime %.1f us; peak at that clock %.1f TFLOP/s; achieved (kernel) %.1f TFLOP/s"
          % (clk / 1e9, dur * 1e6, 157.3 * clk / 2.4e9, fl / dur / 1e12))
