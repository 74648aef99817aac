"""How far is the Python host from being the limit of the training step?  Per-step host time of solver.step() (no sync in the loop)
against the GPU's time per step, then a cProfile of three steps (where the host's time goes).  usage: python scratch/r06/host_probe.py"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver

dev = torch.device("cuda", 0)
net = get_multi_symbol_train("resnet-50", (3, 512, 512), num_classes=8, batch_size=32, device=dev, seed=0)
gen = synthetic.rng(233)
solver = MultiTaskSolver(net)
solver.set_batch(torch.from_numpy(synthetic.images(32, 512, 512, gen)).to(dev),
                 torch.from_numpy(synthetic.det_labels(32, gen=gen, height=512, width=512)).to(dev),
                 torch.from_numpy(synthetic.seg_labels(32, 512, 512, gen=gen)).to(dev))
for _ in range(5):
    solver.step()
torch.cuda.synchronize()
N = 20
host = []
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter()
    solver.step()
    host.append((time.perf_counter() - a) * 1e3)
t_issue = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) * 1e3
print("host time per step (ms):", " ".join("%.1f" % h for h in host))
print("issue of %d steps %.1f ms (%.2f per step); with the final synchronize %.1f ms (%.2f per step)" % (N, t_issue, t_issue / N, t_all, t_all / N))
# the host alone: how long does it take to ISSUE a step when the GPU is far behind?  (same loop, sleeping first so that the queues are empty,
# timing only the first step: the GPU cannot be the limit of its issue)
torch.cuda.synchronize()
a = time.perf_counter(); solver.step(); b = time.perf_counter()
torch.cuda.synchronize()
print("issue time of ONE step into empty queues: %.2f ms" % ((b - a) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    solver.step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
