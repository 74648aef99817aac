"""first end-to-end run of a graph with bf16 activation tensors: finite, deterministic, loss close to the fp32-tensor run"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from dspnet_amd import functional as fn, synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver
from dspnet_amd.train.metric import MultiBoxMetric
net_name = sys.argv[1] if len(sys.argv) > 1 else "resnet-50"
B, S = 2, 256
dev = torch.device("cuda", 0)
res = {}
for store in ("fp32", "bf16"):
    fn.set_activation_dtype(store)
    fn.set_conv_math("bf16")
    net = get_multi_symbol_train(net_name, (3, S, S), num_classes=8, batch_size=B, device=dev, seed=1)
    fn.set_activation_dtype("fp32")
    gen = synthetic.rng(233)
    solver = MultiTaskSolver(net)
    solver.set_batch(torch.from_numpy(synthetic.images(B, S, S, gen)).to(dev),
                     torch.from_numpy(synthetic.det_labels(B, gen=gen, height=S, width=S)).to(dev),
                     torch.from_numpy(synthetic.seg_labels(B, S, S, gen=gen)).to(dev))
    solver.forward(); solver.backward(); torch.cuda.synchronize()
    m = MultiBoxMetric(); m.update(net)
    g1 = net.g.grad_arena.clone()
    solver.forward(); solver.backward(); torch.cuda.synchronize()
    print(store, dict(zip(*m.get())), "finite", bool(torch.isfinite(g1).all()), "deterministic", bool(torch.equal(g1, net.g.grad_arena)),
          "grad norm %.4e" % float(g1.double().norm()))
    res[store] = {p.name: p.grad.clone() for p in net.g.param_order}
    hist = []
    for _ in range(3):
        solver.step(); m.reset(); m.update(net); hist.append(m.get()[1])
    print(store, "3 steps:", hist)
fn.set_conv_math("fp32")
num = den = 0.0
for k, a in res["fp32"].items():
    b = res["bf16"][k]
    if a.shape != b.shape:      # padded channel counts differ (pad4 / pad8)
        if a.dim() == 4: b = b[..., :a.shape[3]]; a = a[..., :b.shape[3]]
        else: n = min(a.numel(), b.numel()); a, b = a[:n], b[:n]
    num += float(((a - b).double() ** 2).sum()); den += float((a.double() ** 2).sum())
print("relative L2 between the gradient sets (fp32 tensors + bf16 math vs bf16 tensors):", (num / den) ** 0.5)
