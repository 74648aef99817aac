import sys
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train, get_det_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver
from dspnet_amd.train.metric import MultiBoxMetric
dev = torch.device("cuda", 0)
for kind, size, B in (("det", 300, 2), ("multi", 256, 2)):
    f = get_det_symbol_train if kind == "det" else get_multi_symbol_train
    net = f("vgg16_reduced", size, num_classes=20 if kind == "det" else 8, batch_size=B, device=dev)
    print(kind, size, "anchors", tuple(net.anchors.shape), "params", net.g.num_params())
    gen = synthetic.rng(1)
    s = MultiTaskSolver(net)
    net.data.data.copy_(torch.from_numpy(synthetic.images(B, size, size, gen)))
    net.label_det.data.copy_(torch.from_numpy(synthetic.det_labels(B, gen=gen, num_classes=8)))
    if net.label_seg is not None:
        net.label_seg.data.copy_(torch.from_numpy(synthetic.seg_labels(B, size, size, gen=gen)))
    for i in range(3):
        s.step()
    torch.cuda.synchronize()
    print([tuple(o.shape) for o in net.outputs()], float(net.g.arena.abs().sum()))
