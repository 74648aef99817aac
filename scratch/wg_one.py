"""one weight-gradient shape under DSPN_WG_TILE / DSPN_WG_SPLITS: python scratch/wg_one.py N H W Cin Cout k"""
import os, subprocess, sys
if sys.argv[1] == "child":
    sys.path.insert(0, '/root/repo')
    import torch
    from dspnet_amd import functional as fn
    N, H, W, Cin, Cout, k = map(int, sys.argv[2:8])
    dev = torch.device("cuda", 0)
    x = torch.randn(N, H, W, Cin, device=dev); dy = torch.randn(N, H, W, (Cout + 3) // 4 * 4, device=dev)
    nsp = fn.L().dspn_conv2d_wgrad_splits(N, H, W, Cin, Cout, k, k, 1)
    slabs = torch.empty(nsp, Cout, k, k, Cin, device=dev)
    f = lambda: fn.conv2d_wgrad_slabs(x, dy, (Cout, k, k, Cin), slabs, 1, k // 2, 1)
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 10
    print("%.3f ms  %.1f TF  (%d splits)" % (t, 2.0 * N * H * W * Cin * Cout * k * k / t / 1e9, nsp))
else:
    for tile in ("",):
        for sp in ("", "4", "7", "14", "28", "56", "112"):
            env = dict(os.environ)
            if tile: env["DSPN_WG_TILE"] = tile
            if sp: env["DSPN_WG_SPLITS"] = sp
            r = subprocess.run([sys.executable, __file__, "child"] + sys.argv[1:7], env=env, capture_output=True, text=True)
            print("tile=%s splits=%s :: %s" % (tile or "auto", sp or "auto", (r.stdout.strip().split("\n") or [""])[-1] or r.stderr[-200:]))
