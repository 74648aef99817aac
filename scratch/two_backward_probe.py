import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from test_graph_gpu import make
net, solver, *_ = make(2, 128, 128)
solver.forward(); solver.backward(); g1 = net.g.grad_arena.clone()
solver.backward(); g2 = net.g.grad_arena.clone()
d = (g2 - g1).abs()
print(net.g.math, "max diff", float(d.max()), "rel", float(d.max() / g1.abs().max()), "nonzero diffs", int((d > 0).sum()), "of", d.numel())
# which params differ
for p in net.g.param_order:
    a = g1[p.offset:p.offset + p.size]; b = g2[p.offset:p.offset + p.size]
    dd = float((a - b).abs().max())
    if dd > 0: print(p.name, dd, float(a.abs().max()), "ratio", float((b.abs().max() / a.abs().max())))
