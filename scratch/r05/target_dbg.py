import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import mbx_cases as mc
from oracle import multibox as om
from dspnet_amd import operator as op
dev = torch.device('cuda', 0)
anc_full = mc.r50_anchors(512, 512)
def run(A, batch, seed, quant, ratio=3.0, max_gt=40):
    anc = anc_full[:, :A].copy()
    lab, pred = mc.target_inputs(anc, batch=batch, seed=seed, max_gt=max_gt)
    if quant:
        pred = torch.from_numpy(pred).to(torch.bfloat16).float().numpy()
        pred = np.round(pred * 4) / 4
    exp = om.multibox_target(anc, lab, pred, negative_mining_ratio=ratio)
    got = op.MultiBoxTarget(torch.from_numpy(anc).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(pred).to(dev), negative_mining_ratio=ratio)
    got = [g.cpu().numpy() for g in got]
    bad = np.argwhere(got[2] != exp[2])
    print("A %5d B %2d seed %d quant %d: cls mismatches %d" % (A, batch, seed, quant, len(bad)), bad[:6].tolist(), [ (got[2][tuple(b)], exp[2][tuple(b)]) for b in bad[:6]])
for A in (3382, 6132, 1000, 4096, 5000):
    for q in (0, 1):
        for seed in (1, 2):
            run(A, 2, seed, q)
