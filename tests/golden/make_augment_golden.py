"""Generates tests/golden/augment_warp.npz from oracle/augment.py (the restatement of OpenCV's warpAffine arithmetic;
no OpenCV is available to produce vectors).  Run from the repo root: python tests/golden/make_augment_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import augment as oa  # noqa: E402

g = np.random.Generator(np.random.PCG64(20261002))
yy, xx = np.mgrid[:24, :40]
img = np.clip(np.stack([120 + 90 * np.sin(xx / (5.0 + c)) * np.cos(yy / (4.0 + c)) for c in range(3)], -1)
              + g.normal(0, 5, (24, 40, 3)), 0, 255).astype(np.uint8)
seg = g.integers(0, 19, (24, 40)).astype(np.uint8)
th = np.radians(3.5)
M = np.array([[1.4 * np.cos(th), -1.2 * np.sin(th), -4.25], [1.4 * np.sin(th), 1.2 * np.cos(th), 2.5]])
out = {"img": img, "seg": seg, "M": M,
       "linear_border128": oa.warp_affine(img, M, (32, 16), True, 128),
       "nearest_border255": oa.warp_affine(seg, M, (32, 16), False, 255)}
hdr = np.array([50, 2, 6] + [3, .2, .25, .6, .7, .4, 1, .05, .1, .3, .35, .2] + [-1.0] * 36)
aug = [1.0, th, 1.4, 1.2, -4.25, 2.5]
h2 = hdr.copy()
oa.get_augmented(np.zeros((24, 40, 3), np.uint8), h2, np.zeros((24, 40), np.uint8), (3, 16, 32), aug)
out["hdr_in"], out["aug"], out["hdr_out"] = hdr, np.array(aug), h2
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "augment_warp.npz"), **out)
print({k: getattr(v, "shape", None) for k, v in out.items()})
