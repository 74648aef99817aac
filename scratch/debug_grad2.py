import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from test_graph_gpu import make
from dspnet_amd.symbol.multitask_symbol_factory import get_config
from oracle import dspnet_torch as ot
import torch.nn.functional as F

net, solver, data, lab, seg = make(2, 256, 256)
g = net.g
solver.forward()
for t in g.tensors.values(): t._gw = False; t.grad = None
# find index of the node producing _plus15
idx15 = [i for i, n in enumerate(g.nodes) if getattr(n, 'out', None) is g.tensors['_plus15']][0]
print("nodes", len(g.nodes), "idx15", idx15)
for idx in range(len(g.nodes) - 1, idx15, -1):
    g.nodes[idx].backward()
torch.cuda.synchronize()
cfg = get_config("resnet-50", 256)
dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(), net.target.cls_target.cpu().numpy()]
keep = {}
orig = ot.resnet50
def patched(P, x):
    inter = orig(P, x)
    for k, v in inter.items(): v.retain_grad(); keep[k] = v
    return inter
ot.resnet50 = patched
origbn = ot.bn
def bn_keep(x, gamma, beta, relu=False):
    y = origbn(x, gamma, beta, relu)
    if x.requires_grad and x.shape[1] == 2048 and gamma is None:
        y.retain_grad(); keep['r5'] = y
    return y
ot.bn = bn_keep
ref = ot.forward_loss(ot.export_params(g), data, lab, seg, cfg["sizes"][1:], cfg["ratios"][1:], dtype=torch.float64, targets=dev_targets)
ref["objective"].backward()
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
def dev(name): return g.tensors[name].grad.cpu().numpy().transpose(0, 3, 1, 2)
print("G15", rel(dev('_plus15'), keep['_plus15'].grad.numpy()))
print("r5 grad", rel(dev('res5_reduced_bn_out'), keep['r5'].grad.numpy()))
# partial G12: only heads + stage4 not run -> compare only head contribution is impossible; print norms
print("G15 norms", np.abs(dev('_plus15')).max(), np.abs(keep['_plus15'].grad.numpy()).max())
d = dev('_plus15') - keep['_plus15'].grad.numpy()
print("diff per-sample max", np.abs(d).reshape(2, -1).max(1), "channel argmax", np.unravel_index(np.abs(d).argmax(), d.shape))
