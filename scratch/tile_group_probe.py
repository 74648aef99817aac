"""bn_stats_from_tiles (tile_group + final) over the tile-table shapes of the graph"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from dspnet_amd import functional as fn
dev = torch.device("cuda", 0)
def timeit(f, reps=20):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for rows, C in [(2097152, 64), (524288, 64), (524288, 256), (131072, 512), (131072, 128), (32768, 1024)]:
    tiles, tile_rows = fn.conv_stats_layout(rows, C)
    ts = torch.randn(tiles, 2, C, device=dev).abs()
    v = [torch.zeros(C, device=dev) for _ in range(6)]
    t = timeit(lambda: fn.bn_stats_from_tiles(ts, tiles, tile_rows, rows, C, 2e-5, v[0], v[1], v[2], v[3], v[4], v[5]))
    print("rows %8d C %4d tiles %6d x %3d rows: %.1f us" % (rows, C, tiles, tile_rows, t))
