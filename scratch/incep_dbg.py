import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from dspnet_amd import synthetic
from dspnet_amd.symbol.multitask_symbol_factory import get_config, get_multi_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver
from oracle import dspnet_torch as ot
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
size = 512
dev = torch.device("cuda", 0)
net = get_multi_symbol_train("inceptionv3", size, num_classes=8, batch_size=B, device=dev, seed=3)
gen = synthetic.rng(77)
data = synthetic.images(B, size, size, gen)
lab = synthetic.det_labels(B, gen=gen, num_classes=8, height=size, width=size, first_empty=False)
seg = synthetic.seg_labels(B, size, size, gen=gen)
net.data.data.copy_(torch.from_numpy(data).to(dev)); net.label_det.data.copy_(torch.from_numpy(lab).to(dev)); net.label_seg.data.copy_(torch.from_numpy(seg).to(dev))
s = MultiTaskSolver(net); s.forward(); torch.cuda.synchronize()
cfg = get_config("inceptionv3", size)
tg = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(), net.target.cls_target.cpu().numpy()]
vals = ot.export_params(net.g)
r64 = ot.forward_loss(vals, data, lab, seg, num_classes=8, dtype=torch.float64, targets=tg, config=cfg)
r32 = ot.forward_loss(vals, data, lab, seg, num_classes=8, dtype=torch.float32, targets=tg, config=cfg)
def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
d = net.loc_preds.data.cpu().numpy()
print("B", B, "dev vs f64", rel(d, r64["loc_preds"].numpy()), "cpu f32 vs f64", rel(r32["loc_preds"].numpy(), r64["loc_preds"].numpy()))
print("seg dev vs f64", rel(net.outputs()[4].cpu().numpy(), r64["seg_out"].numpy()), "cpu f32 vs f64", rel(r32["seg_out"].numpy(), r64["seg_out"].numpy()))
for k in ("CrossEntropy", "SmoothL1", "SegCrossEntropy"):
    print(k, r64[k], r32[k])
s.backward(); torch.cuda.synchronize()
r64["objective"].backward()
rows = []
num = den = 0
for p in net.g.param_order:
    gref = ot.import_grad(p.name, r64["params"][p.name].grad)
    gdev = p.grad.cpu().numpy()
    gdev = gdev[:gref.shape[0], :, :, :gref.shape[3]] if gdev.ndim == 4 else gdev[:gref.shape[0]]
    e = float(((gdev - gref) ** 2).sum()); d = float((gref ** 2).sum())
    num += e; den += d
    rows.append(((e / (d + 1e-300)) ** 0.5, p.name, d ** 0.5))
print("global", (num / den) ** 0.5)
for r in rows: print("%.3e %-60s |g|=%.3e" % r)
r32["objective"].backward()
num = den = 0
for p in net.g.param_order:
    a = r32["params"][p.name].grad.double().numpy(); b = r64["params"][p.name].grad.numpy()
    num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum())
print("CPU f32 vs f64 global grad rel L2", (num / den) ** 0.5)
for nm in ("mixed_10_conv_conv2d_weight", "mixed_7_conv_batchnorm_beta", "conv_3_batchnorm_beta", "multi_feat_1_conv_1x1_conv_weight"):
    a = r32["params"][nm].grad.double().numpy(); b = r64["params"][nm].grad.numpy()
    print(nm, float((((a - b) ** 2).sum() / (b ** 2).sum()) ** 0.5))
