import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import mbx_cases as mc
from oracle import multibox as om
from dspnet_amd import operator as op, synthetic, functional as fn
from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
from dspnet_amd.train.solver import MultiTaskSolver
fn.set_conv_math("bf16")
dev = torch.device("cuda", 0)
H, W, B = 512, 1024, 1
net = get_multi_symbol_train("inceptionv3", (3, H, W), num_classes=8, batch_size=B, device=dev, seed=3)
gen = synthetic.rng(78)
data = synthetic.images(B, H, W, gen)
lab = synthetic.det_labels(B, gen=gen, height=H, width=W, first_empty=False)
seg = synthetic.seg_labels(B, H, W, gen=gen)
solver = MultiTaskSolver(net)
solver.set_batch(torch.from_numpy(data).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(seg).to(dev))
solver.forward(); torch.cuda.synchronize()
anchors = net.anchors.cpu().numpy()
pred = net.target.cls_preds.data.cpu().numpy()
exp = om.multibox_target(anchors, lab, pred, negative_mining_ratio=3)
got = [net.target.loc_target.cpu().numpy(), net.target.loc_mask.cpu().numpy(), net.target.cls_target.cpu().numpy()]
bad = np.argwhere(got[2] != exp[2])
G = int((lab[0, :, 0] >= 0).sum())
print("A", anchors.shape, "G", G, "mismatches", bad.tolist(), [(got[2][tuple(b)], exp[2][tuple(b)]) for b in bad])
# standalone operator on the same inputs
got2 = op.MultiBoxTarget(torch.from_numpy(anchors).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(pred).to(dev), negative_mining_ratio=3)
bad2 = np.argwhere(got2[2].cpu().numpy() != exp[2])
print("standalone mismatches", bad2.tolist())
def iou(a, g):
    iw = max(0, min(a[2], g[2]) - max(a[0], g[0])); ih = max(0, min(a[3], g[3]) - max(a[1], g[1]))
    i = iw * ih; u = (a[2]-a[0])*(a[3]-a[1]) + (g[2]-g[0])*(g[3]-g[1]) - i
    return i / u if u else 0
for b in bad[:8]:
    j = b[1]
    ious = [iou(anchors[0, j], lab[0, k, 1:5]) for k in range(G)]
    print("anchor", j, anchors[0, j], "best iou", max(ious), "argmax", int(np.argmax(ious)))
for k in range(G):
    ious = np.array([iou(anchors[0, j], lab[0, k, 1:5]) for j in range(anchors.shape[1])])
    m = ious.max(); idx = np.flatnonzero(ious == m)
    print("gt", k, "max iou %.7f" % m, "at", idx[:6].tolist())
