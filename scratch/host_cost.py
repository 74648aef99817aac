"""host-side cost of one training step: CPU seconds (process_time, all threads) and wall time per step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver
dev = torch.device("cuda", 0)
B, S = (int(os.environ.get("HB", 32)), int(os.environ.get("HS", 512)))
net = get_multi_symbol_train("resnet-50", S, num_classes=8, batch_size=B, device=dev)
solver = MultiTaskSolver(net)
gen = synthetic.rng(233)
solver.set_batch(torch.from_numpy(synthetic.images(B, S, S, gen)).to(dev),
                 torch.from_numpy(synthetic.det_labels(B, gen=gen, height=S, width=S)).to(dev),
                 torch.from_numpy(synthetic.seg_labels(B, S, S, gen=gen)).to(dev))
for _ in range(3): solver.step()
torch.cuda.synchronize()
K = 20
c0, t0, th0 = time.process_time(), time.perf_counter(), time.thread_time()
for _ in range(K): solver.step()
c1, t1, th1 = time.process_time(), time.perf_counter(), time.thread_time()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("per step: enqueue wall %.1f ms, main-thread CPU %.1f ms, process CPU (all threads) %.1f ms, wall incl. GPU %.1f ms; cores allowed %s"
      % ((t1 - t0) / K * 1e3, (th1 - th0) / K * 1e3, (c1 - c0) / K * 1e3, (t2 - t0) / K * 1e3, len(os.sched_getaffinity(0))))
