"""per-parameter gradient error of the segmentation-only graph against the float64 restatement"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from dspnet_amd.symbol.multitask_symbol_factory import get_seg_symbol_train, get_config
from dspnet_amd.train.solver import MultiTaskSolver
from dspnet_amd import synthetic
from oracle import dspnet_torch as ot
dev = torch.device("cuda", 0)
B, S = 2, 256
net = get_seg_symbol_train("resnet-50", S, num_classes=8, batch_size=B, device=dev, seed=5)
gen = synthetic.rng(31)
data = synthetic.images(B, S, S, gen); seg = synthetic.seg_labels(B, S, S, gen=gen)
solver = MultiTaskSolver(net)
solver.set_batch(torch.from_numpy(data).to(dev), None, torch.from_numpy(seg).to(dev))
solver.forward(); solver.backward(); torch.cuda.synchronize()
ref = ot.forward_loss(ot.export_params(net.g), data, None, seg, num_classes=8, dtype=torch.float64,
                      config=get_config("resnet-50", S), with_det=False)
ref["objective"].backward()
rows = []
for p in net.g.param_order:
    g = ref["params"][p.name].grad
    if g is None:
        print("no ref grad", p.name, float(p.grad.abs().max())); continue
    gref = ot.import_grad(p.name, g); gdev = p.grad.cpu().numpy()
    gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]]
    rows.append((float(np.sqrt(((gdev - gref) ** 2).sum())), float(np.sqrt((gref ** 2).sum())), p.name))
rows.sort(reverse=True)
for e, n, name in rows[:25]:
    print("%-40s err %.3e norm %.3e rel %.2e" % (name, e, n, e / (n + 1e-30)))
