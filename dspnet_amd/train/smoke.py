"""One tiny forward+backward+update of the multi-task graph on cuda:0 (used by __graft_entry__.smoke)."""
import numpy as np
import torch

from .. import synthetic
from ..symbol.multitask_symbol_factory import get_multi_symbol_train
from .metric import MultiBoxMetric
from .solver import MultiTaskSolver


def run(batch=2, size=128):
    dev = torch.device("cuda", 0)
    net = get_multi_symbol_train("resnet-50", size, num_classes=8, batch_size=batch, device=dev)
    gen = synthetic.rng(233)
    solver = MultiTaskSolver(net)
    solver.set_batch(torch.from_numpy(synthetic.images(batch, size, size, gen)).to(dev),
                     torch.from_numpy(synthetic.det_labels(batch, gen=gen, height=size, width=size)).to(dev),
                     torch.from_numpy(synthetic.seg_labels(batch, size, size, gen=gen)).to(dev))
    m = MultiBoxMetric()
    losses = []
    for _ in range(3):
        solver.step()
        m.reset(); m.update(net)
        losses.append(m.get()[1])
    torch.cuda.synchronize()
    arr = np.asarray(losses)
    assert np.isfinite(arr).all(), arr
    outs = net.outputs()
    assert outs[3].shape[2] == 7 and outs[4].shape[1] == 19
    print("train smoke losses", arr.tolist())
    return arr
