import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from dspnet_amd import engine as E, functional as fn
from test_graph_gpu import make
from dspnet_amd.symbol.multitask_symbol_factory import get_config
from oracle import dspnet_torch as ot
net, solver, data, lab, seg = make(2, 256, 256)
g = net.g
solver.forward(); solver.backward(); torch.cuda.synchronize()
cfg = get_config("resnet-50", 256)
dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(), net.target.cls_target.cpu().numpy()]
outs = []
origbn = ot.bn
def bn_keep(x, gamma, beta, relu=False):
    y = origbn(x, gamma, beta, relu)
    if y.requires_grad: y.retain_grad()
    outs.append(y)
    return y
ot.bn = bn_keep
ref = ot.forward_loss(ot.export_params(g), data, lab, seg, cfg["sizes"][1:], cfg["ratios"][1:], dtype=torch.float64, targets=dev_targets)
ref["objective"].backward()
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
bns = [n for n in g.nodes if isinstance(n, E.BatchNorm)]
print(len(bns), len(outs))
for n, o in list(zip(bns, outs))[40:53]:
    name = n.out.name
    gd = n.out.grad
    if gd is None or o.grad is None: print(name, "none"); continue
    gd = gd.cpu().numpy().transpose(0, 3, 1, 2)[:, :o.shape[1]]
    fd = n.out.data.cpu().numpy().transpose(0, 3, 1, 2)[:, :o.shape[1]]
    print("%-28s fwd %.2e grad %.2e  |g| %.3e" % (name, rel(fd, o.detach().numpy()), rel(gd, o.grad.numpy()), np.abs(o.grad.numpy()).max()))
