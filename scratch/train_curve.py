"""loss trajectory of repeated steps on one synthetic batch (sanity of the whole forward / backward / update chain)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.metric import MultiBoxMetric
from dspnet_amd.train.solver import MultiTaskSolver
dev = torch.device("cuda", 0)
B, S = 8, 256
net = get_multi_symbol_train(sys.argv[1] if len(sys.argv) > 1 else "resnet-50", S, num_classes=8, batch_size=B, device=dev)
solver = MultiTaskSolver(net, learning_rate=0.002)
gen = synthetic.rng(233)
solver.set_batch(torch.from_numpy(synthetic.images(B, S, S, gen)).to(dev),
                 torch.from_numpy(synthetic.det_labels(B, gen=gen, height=S, width=S)).to(dev),
                 torch.from_numpy(synthetic.seg_labels(B, S, S, gen=gen)).to(dev))
m = MultiBoxMetric()
for step in range(201):
    solver.step()
    if step % 25 == 0:
        m.reset(); m.update(net)
        print(step, ["%s %.4f" % kv for kv in zip(*m.get())])
