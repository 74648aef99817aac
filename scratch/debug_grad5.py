import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from dspnet_amd import engine as E, functional as fn
from test_graph_gpu import make
net, solver, data, lab, seg = make(2, 256, 256)
g = net.g
solver.forward(); solver.backward(); torch.cuda.synchronize()
n = [n for n in g.nodes if isinstance(n, E.BatchNorm) and n.out.name == "stage4_unit3_bn1_relu"][0]
x, y, dy = n.x.data, n.out.data, n.out.grad
dx, dgam, dbet = fn.bn_backward(x, y, dy, n.mean, n.rstd, n.gamma.data, relu=True)
torch.cuda.synchronize()
X, Y, DY = x.cpu().double().reshape(-1, 2048), y.cpu().double().reshape(-1, 2048), dy.cpu().double().reshape(-1, 2048)
g_ = DY * (Y > 0)
S = g_.sum(0)
xh = (X - n.mean.cpu().double()) * n.rstd.cpu().double()
SS = (g_ * xh).sum(0)
def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-30))
print("standalone dbeta vs host", rel(dbet.cpu().double(), S), "dgamma", rel(dgam.cpu().double(), SS))
print("graph dbeta vs host", rel(n.beta.grad.cpu().double(), S), "graph dgamma", rel(n.gamma.grad.cpu().double(), SS))
print("|S| max", float(S.abs().max()), "sum|g| max", float(g_.abs().sum(0).max()))
