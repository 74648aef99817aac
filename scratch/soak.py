"""300 SGD steps on one fixed synthetic batch (resnet-50 multi-task, 512x512, bs 8) in the split math and on the fp32 MFMA:
the losses must stay finite and fall; prints every 50th step"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from dspnet_amd import functional as fn, synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.metric import MultiBoxMetric
from dspnet_amd.train.solver import MultiTaskSolver
dev = torch.device("cuda", 0)
for math in ("f16x2", "bf16x3", "fp32"):
    fn.set_conv_math(math)
    net = get_multi_symbol_train("resnet-50", 512, num_classes=8, batch_size=8, device=dev, seed=0)
    g = synthetic.rng(3)
    solver = MultiTaskSolver(net)
    solver.set_batch(torch.from_numpy(synthetic.images(8, 512, 512, g)).to(dev),
                     torch.from_numpy(synthetic.det_labels(8, gen=g, height=512, width=512, first_empty=False)).to(dev),
                     torch.from_numpy(synthetic.seg_labels(8, 512, 512, gen=g)).to(dev))
    m = MultiBoxMetric()
    for step in range(300):
        solver.step()
        if step % 50 == 0 or step == 299:
            m.reset(); m.update(net)
            print(math, step, ["%.4f" % v for v in m.get()[1]], "finite" if bool(torch.isfinite(net.g.arena).all()) else "NON-FINITE")
