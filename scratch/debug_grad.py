import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from test_graph_gpu import make
from dspnet_amd.symbol.multitask_symbol_factory import get_config
from oracle import dspnet_torch as ot

net, solver, data, lab, seg = make(2, 256, 256)
solver.forward(); solver.backward(); torch.cuda.synchronize()
cfg = get_config("resnet-50", 256)
dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(), net.target.cls_target.cpu().numpy()]
# monkeypatch oracle to retain grads of internals
orig = ot.resnet50
keep = {}
def patched(P, x):
    inter = orig(P, x)
    for k, v in inter.items():
        v.retain_grad(); keep[k] = v
    return inter
ot.resnet50 = patched
ref = ot.forward_loss(ot.export_params(net.g), data, lab, seg, cfg["sizes"][1:], cfg["ratios"][1:], dtype=torch.float64, targets=dev_targets)
ref["objective"].backward()
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
for name in ["_plus13", "_plus7", "_plus3", "_plus0"]:
    gd = net.g.tensors[name].grad.cpu().numpy().transpose(0, 3, 1, 2)
    gr = keep[name].grad.numpy()
    print(name, "grad rel err", rel(gd, gr), "fwd rel err", rel(net.g.tensors[name].data.cpu().numpy().transpose(0,3,1,2), keep[name].detach().numpy()))
errs = {}
for p in net.g.param_order:
    gref = ot.import_grad(p.name, ref["params"][p.name].grad)
    gdev = p.grad.cpu().numpy()
    gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]]
    errs[p.name] = rel(gdev, gref)
for k, v in errs.items():
    if v > 1e-3: print("%-40s %.4f" % (k, v))
