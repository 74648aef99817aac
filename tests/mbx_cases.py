"""Shared builders for multibox test cases (numpy only)."""
import numpy as np

from dspnet_amd import synthetic
from oracle import multibox as om

# resnet-50 preset after the [1:] slice (symbol/multitask_symbol_factory.py:68-81,
# symbol/multitask_symbol_builder.py:503-508)
R50_SIZES = [[.1, .141], [.2, .272], [.37, .447], [.54, .619], [.71, .79], [.88, .961]]
R50_RATIOS = [[1, 2, .5], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3],
              [1, 2, .5], [1, 2, .5]]


def r50_maps(height, width):
    maps = []
    for s in (16, 32):
        maps.append((height // s, width // s))
    h, w = maps[-1]
    for _ in range(4):  # 3x3 stride-2 pad-1 extras
        h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        maps.append((h, w))
    return maps


def r50_anchors(height=512, width=512):
    parts = [om.multibox_prior(h, w, s, r)
             for (h, w), s, r in zip(r50_maps(height, width), R50_SIZES, R50_RATIOS)]
    return np.concatenate(parts, axis=1)


def small_anchors(h=4, w=5):
    return np.concatenate([om.multibox_prior(h, w, [.2, .3], [1, 2, .5]),
                           om.multibox_prior(2, 2, [.5, .7], [1, 2])], axis=1)


def target_inputs(anchors, batch, num_labels=200, num_classes=8, max_gt=40, seed=233,
                  pred_scale=1.0):
    gen = synthetic.rng(seed)
    lab = synthetic.det_labels(batch, num_labels, num_classes, max_gt, gen=gen)
    A = anchors.shape[1]
    cls_pred = (gen.standard_normal((batch, num_classes + 1, A)) * pred_scale).astype(np.float32)
    return lab, cls_pred


def detection_inputs(anchors, batch, num_classes=8, seed=7, peaky=True):
    gen = synthetic.rng(seed)
    A = anchors.shape[1]
    logits = gen.standard_normal((batch, num_classes + 1, A)).astype(np.float32)
    if peaky:  # make background dominant for most anchors, like a trained net
        logits[:, 0, :] += 3.0
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    prob = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    loc = (gen.standard_normal((batch, A * 5)) * 0.5).astype(np.float32)
    return prob, loc


def assert_target_equal(got, exp):
    """[loc_target, loc_mask, cls_target]: masks / classes / which anchors are positive must be
    bit-exact; so must dx, dy and dist.  The two log() columns depend on libm's logf (0.818 ulp,
    not reproducible off-host): 1-ulp tolerance there."""
    lt_g, lm_g, ct_g = [np.asarray(x) for x in got]
    lt_e, lm_e, ct_e = [np.asarray(x) for x in exp]
    np.testing.assert_array_equal(ct_g, ct_e)
    np.testing.assert_array_equal(lm_g, lm_e)
    B = lt_g.shape[0]
    g5, e5 = lt_g.reshape(B, -1, 5), lt_e.reshape(B, -1, 5)
    np.testing.assert_array_equal(g5[..., [0, 1, 4]], e5[..., [0, 1, 4]])
    np.testing.assert_allclose(g5[..., 2:4], e5[..., 2:4], rtol=2.5e-7, atol=1e-7)
