"""wgrad of the stage-1 1x1 layers under forced tile / split settings (DSPN_WG_TILE, DSPN_WG_SPLITS), one process per setting"""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, '/root/repo')
    import torch
    from dspnet_amd import functional as fn
    dev = torch.device("cuda", 0)
    def timeit(f, reps=10):
        f(); f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    out = []
    for (N, H, W, Cin, Cout, k) in [(32, 128, 128, 64, 64, 1), (32, 128, 128, 64, 256, 1), (32, 128, 128, 256, 64, 1), (32, 64, 64, 256, 128, 1)]:
        x = torch.randn(N, H, W, Cin, device=dev); dy = torch.randn(N, H, W, Cout, device=dev)
        sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev)
        t = timeit(lambda: fn.conv2d_wgrad(x, dy, (Cout, k, k, Cin), 1, 0, 1))
        nsp = fn.L().dspn_conv2d_wgrad_splits(N, H, W, Cin, Cout, k, k, 1)
        slabs = torch.empty(nsp, Cout, k, k, Cin, device=dev)
        t2 = timeit(lambda: fn.conv2d_wgrad_slabs(x, dy, (Cout, k, k, Cin), slabs, 1, 0, 1))   # GEMM alone
        gb = 4.0 * N * H * W * (Cin + Cout) / 1e9
        out.append("%d->%d %.3f/%.3f ms (%d sp, %.1f TB/s)" % (Cin, Cout, t, t2, nsp, gb / t2))
    print(" | ".join(out))
else:
    for tile in ("", "0", "1", "2"):
        for sp in ("", "48", "96", "192", "384", "768"):
            env = dict(os.environ)
            if tile: env["DSPN_WG_TILE"] = tile
            if sp: env["DSPN_WG_SPLITS"] = sp
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print("tile=%s splits=%s :: %s" % (tile or "auto", sp or "auto", r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-300:]))
