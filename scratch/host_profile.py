"""cProfile of the host side of the training step (where the enqueue time goes)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver
dev = torch.device("cuda", 0)
B, S = 32, 512
net = get_multi_symbol_train("resnet-50", S, num_classes=8, batch_size=B, device=dev)
solver = MultiTaskSolver(net)
gen = synthetic.rng(233)
solver.set_batch(torch.from_numpy(synthetic.images(B, S, S, gen)).to(dev),
                 torch.from_numpy(synthetic.det_labels(B, gen=gen, height=S, width=S)).to(dev),
                 torch.from_numpy(synthetic.seg_labels(B, S, S, gen=gen)).to(dev))
for _ in range(3): solver.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): solver.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(22)
