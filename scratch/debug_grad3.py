import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from dspnet_amd import engine as E, functional as fn
MODE = sys.argv[1]
if MODE == "copy":
    def give_grad(self, buf):
        if not self._gw:
            self.grad = self.own_grad(); self.grad.copy_(buf); self._gw = True
        else:
            fn.add(self.grad, buf, out=self.grad)
    E.Tensor.give_grad = give_grad
from test_graph_gpu import make
from dspnet_amd.symbol.multitask_symbol_factory import get_config
from oracle import dspnet_torch as ot
net, solver, data, lab, seg = make(2, 256, 256)
g = net.g
solver.forward(); solver.backward(); torch.cuda.synchronize()
cfg = get_config("resnet-50", 256)
dev_targets = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(), net.target.cls_target.cpu().numpy()]
ref = ot.forward_loss(ot.export_params(g), data, lab, seg, cfg["sizes"][1:], cfg["ratios"][1:], dtype=torch.float64, targets=dev_targets)
ref["objective"].backward()
def rel(a, b): return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))
errs = {}
for p in g.param_order:
    gref = ot.import_grad(p.name, ref["params"][p.name].grad)
    gdev = p.grad.cpu().numpy()
    gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]]
    errs[p.name] = rel(gdev, gref)
bad = {k: v for k, v in errs.items() if v > 1e-4}
print(MODE, "num bad", len(bad), sorted(bad.items(), key=lambda kv: -kv[1])[:5])
for k in ["stage4_unit3_conv3_weight", "stage4_unit3_bn3_beta", "stage4_unit3_conv2_weight", "stage4_unit3_bn2_beta", "stage4_unit3_conv1_weight", "stage4_unit3_bn1_beta", "stage4_unit3_bn1_gamma", "stage4_unit2_conv3_weight"]:
    print(k, errs[k])
