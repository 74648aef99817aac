"""Generates tests/golden/multibox_small.npz from the C oracle (inputs + expected outputs).

The reference cannot run here (Python 2 + MXNet), so these vectors pin the
ORACLE, not the reference: they make oracle regressions visible and give the GPU tests a
fixed, committed target.  Run from the repo root: python tests/golden/make_multibox_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mbx_cases as mc  # noqa: E402
from oracle import multibox as om  # noqa: E402

anc = mc.small_anchors(8, 9)
lab, pred = mc.target_inputs(anc, batch=4, num_labels=16, num_classes=5, max_gt=9, seed=11)
lt, lm, ct = om.multibox_target(anc, lab, pred, negative_mining_ratio=3, negative_mining_thresh=.5,
                                overlap_threshold=.5)
prob, loc = mc.detection_inputs(anc, batch=4, num_classes=5, seed=12, peaky=False)
det = om.multibox_detection(prob, loc, anc, nms_threshold=.45, nms_topk=20)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "multibox_small.npz"),
                    anchors=anc, label=lab, cls_pred=pred, loc_target=lt, loc_mask=lm,
                    cls_target=ct, cls_prob=prob, loc_pred=loc, det=det)
print("wrote multibox_small.npz", anc.shape, int((ct > 0).sum()), int((det[..., 0] >= 0).sum()))
