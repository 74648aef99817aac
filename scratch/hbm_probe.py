"""what a plain streaming kernel reaches on this box: copy, read-only sum, write-only fill (torch ops), 1 GiB tensors"""
import torch
dev = torch.device("cuda", 0)
n = 256 * 1024 * 1024
x = torch.randn(n, device=dev); y = torch.empty_like(x)
def timeit(f, reps=10):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t = timeit(lambda: y.copy_(x)); print("copy 1 GiB -> 1 GiB: %.3f ms  %.2f TB/s (read+write)" % (t, 2 * n * 4 / t / 1e9))
t = timeit(lambda: x.sum()); print("sum of 1 GiB: %.3f ms  %.2f TB/s" % (t, n * 4 / t / 1e9))
t = timeit(lambda: y.fill_(1.0)); print("fill 1 GiB: %.3f ms  %.2f TB/s" % (t, n * 4 / t / 1e9))
t = timeit(lambda: torch.add(x, 1.0, out=y)); print("y = x + 1: %.3f ms  %.2f TB/s" % (t, 2 * n * 4 / t / 1e9))
