"""GPU parity of the HIP multibox operators (through the C ABI) against the CPU oracle.

Bar: bit-exact for every index-like output (which anchors are positive / negative / ignored,
class targets, masks, detection ids, row order, suppression) and for every float that does not
pass through libm's logf; 1 ulp on the two log() columns of loc_target."""
import os

import numpy as np
import pytest
import torch

import mbx_cases as mc
from dspnet_amd import operator as op
from dspnet_amd._lib import DspnError
from oracle import multibox as om

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(ts):
    return [t.cpu().numpy() for t in ts]


# ---------------------------------------------------------------- prior
@pytest.mark.parametrize("h,w", [(1, 1), (2, 2), (3, 7), (32, 32), (64, 128)])
@pytest.mark.parametrize("clip", [False, True])
def test_prior_bit_exact(gpu_device, h, w, clip):
    sizes, ratios = [.1, .141], [1, 2, .5, 3, 1. / 3]
    got = op.MultiBoxPrior(torch.empty(2, 8, h, w, device=gpu_device), sizes=sizes, ratios=ratios,
                           clip=clip).cpu().numpy()
    np.testing.assert_array_equal(got, om.multibox_prior(h, w, sizes, ratios, clip=clip))


def test_prior_steps_offsets_and_string_attrs(gpu_device):
    got = op.MultiBoxPrior((5, 9), sizes="(0.3,0.5,0.7)", ratios="(1,2)", steps=(0.2, 0.1),
                           offsets=(0.25, 0.75)).cpu().numpy()
    exp = om.multibox_prior(5, 9, [.3, .5, .7], [1, 2], steps=(0.2, 0.1), offsets=(0.25, 0.75))
    np.testing.assert_array_equal(got, exp)


def test_r50_anchor_table(gpu_device):
    parts = [op.MultiBoxPrior((h, w), sizes=s, ratios=r)
             for (h, w), s, r in zip(mc.r50_maps(512, 512), mc.R50_SIZES, mc.R50_RATIOS)]
    got = torch.cat(parts, dim=1).cpu().numpy()
    np.testing.assert_array_equal(got, mc.r50_anchors(512, 512))


# ---------------------------------------------------------------- target
def run_target(anc, lab, pred, **kw):
    got = host(op.MultiBoxTarget(dev(anc), dev(lab), dev(pred), **kw))
    exp = om.multibox_target(anc, lab, pred, **kw)
    mc.assert_target_equal(got, exp)
    return got


def test_target_self_generated_regression_fixture(gpu_device):
    g = np.load(os.path.join(GOLDEN, "multibox_small.npz"))
    got = host(op.MultiBoxTarget(dev(g["anchors"]), dev(g["label"]), dev(g["cls_pred"]),
                                 negative_mining_ratio=3, negative_mining_thresh=.5,
                                 overlap_threshold=.5))
    mc.assert_target_equal(got, [g["loc_target"], g["loc_mask"], g["cls_target"]])


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("ratio,thr,nthr", [(3.0, 0.5, 0.5), (-1.0, 0.5, 0.5), (3.0, 0.0, 0.5),
                                            (1.5, 0.3, 0.2), (0.4, 0.6, 0.9)])
def test_target_small_random(gpu_device, seed, ratio, thr, nthr):
    anc = mc.small_anchors(6, 7)
    lab, pred = mc.target_inputs(anc, batch=5, num_labels=12, num_classes=4, max_gt=8, seed=seed)
    run_target(anc, lab, pred, overlap_threshold=thr, negative_mining_ratio=ratio,
               negative_mining_thresh=nthr)


@pytest.mark.parametrize("hw,batch", [((512, 512), 8), ((512, 1024), 3), ((512, 512), 32)])
def test_target_dspnet_shape(gpu_device, hw, batch):
    """the training configuration of symbol/multitask_symbol_builder.py:517-521 (round 4: also at the headline batch, 32 x 6132
    anchors, against the oracle -- the greedy matching loop of target_match_kernel was rewritten this round)"""
    anc = mc.r50_anchors(*hw)
    lab, pred = mc.target_inputs(anc, batch=batch, seed=233)
    got = run_target(anc, lab, pred, overlap_threshold=.5, ignore_label=-1,
                     negative_mining_ratio=3, minimum_negative_samples=0,
                     negative_mining_thresh=.5, variances=(0.1, 0.1, 0.2, 0.2))
    ct = got[2]
    assert (ct[-1] == -1).all()                    # the sample generated with zero GT rows
    npos, nneg = (ct > 0).sum(axis=1), (ct == 0).sum(axis=1)
    assert (nneg[:-1] == 3 * npos[:-1]).all() and npos[:-1].min() >= 1


def test_target_all_ties_resolved_by_index(gpu_device):
    anc = mc.r50_anchors(512, 512)
    lab, _ = mc.target_inputs(anc, batch=4, seed=5)
    pred = np.zeros((4, 9, anc.shape[1]), np.float32)      # what a zero-initialised head emits
    run_target(anc, lab, pred, negative_mining_ratio=3)


def test_target_duplicate_and_overlapping_gt(gpu_device):
    """several GTs share their best anchor -> the column-maximum recomputation path"""
    anc = mc.r50_anchors(512, 512)
    lab = -np.ones((3, 200, 6), np.float32)
    for k in range(12):
        lab[0, k] = [k % 8, .30, .30, .62, .62, .1 * (k % 9)]            # identical boxes
    for k in range(30):
        lab[1, k] = [k % 8, .2 + .002 * k, .2, .5 + .002 * k, .5, .5]    # near-duplicates
    lab[2, 0] = [0, .0, .0, .004, .004, .3]                              # tiny box, IoU ~ 0
    lab[2, 1] = [1, .9, .9, .9, .9, .3]                                  # zero-area box
    gen = np.random.default_rng(3)
    pred = gen.standard_normal((3, 9, anc.shape[1])).astype(np.float32)
    run_target(anc, lab, pred, negative_mining_ratio=3)


@pytest.mark.parametrize("num_anchors", [3382, 256 * 5 + 7, 256 * 9 + 33, 64 * 3 + 50, 6132])
def test_target_best_anchor_in_the_last_partial_wave(gpu_device, num_anchors):
    """round 5: the best anchor of a ground truth is found per 64-anchor wave inside target_rows_kernel (a butterfly over the
    wave's lanes); ground truths that coincide with the LAST anchors of the table have their column maximum in a wave whose
    upper lanes hold no anchor (the inceptionv3 1024x512 graph, 3382 anchors, met this case)"""
    anc = mc.r50_anchors(512, 512)[:, :num_anchors].copy()
    lab = -np.ones((2, 40, 6), np.float32)
    for k, j in enumerate(range(num_anchors - 1, num_anchors - 12, -2)):
        box = np.clip(anc[0, j], 0.0, 1.0)
        lab[0, k] = [k % 8, box[0], box[1], box[2], box[3], .1 * k]
    lab[1, 0] = [3, *np.clip(anc[0, num_anchors - 3], 0.0, 1.0), .5]
    lab[1, 1] = [5, .1, .1, .4, .5, .5]
    gen = np.random.default_rng(11)
    pred = gen.standard_normal((2, 9, num_anchors)).astype(np.float32)
    run_target(anc, lab, pred, negative_mining_ratio=3)


def test_target_full_label_table_and_many_positives(gpu_device):
    anc = mc.small_anchors(10, 10)
    lab, pred = mc.target_inputs(anc, batch=2, num_labels=7, num_classes=3, max_gt=7, seed=8)
    lab[0] = mc.target_inputs(anc, batch=1, num_labels=7, num_classes=3, max_gt=7, seed=21)[0][0]
    lab[0, :, 0] = np.abs(lab[0, :, 0])          # no -1 terminator at all: G == L
    lab[0, :, 1:5] = np.abs(lab[0, :, 1:5])
    run_target(anc, lab, pred, negative_mining_ratio=3, overlap_threshold=0.1)


def test_target_reference_aborts_become_codes(gpu_device):
    anc = mc.small_anchors()
    lab = -np.ones((2, 4, 6), np.float32)
    lab[0, 0] = [0, .1, .1, .5, .5, .3]
    lab[1, 0] = [0, .1, .1, .5, .5, .3]
    lab[1, 1] = [-1, .2, -1, -1, -1, -1]
    pred = np.zeros((2, 3, anc.shape[1]), np.float32)
    op.MultiBoxTarget(dev(anc), dev(lab), dev(pred), negative_mining_ratio=3)   # async: no raise
    with pytest.raises(DspnError, match="padded label row"):                     # ... until the deferred check
        op.MultiBoxTarget_check(2, dev(anc).device)
    with pytest.raises(DspnError, match="padded label row"):
        op.MultiBoxTarget(dev(anc), dev(lab), dev(pred), negative_mining_ratio=3, check_errors=True)
    _, rc = om.multibox_target(anc, lab, pred, negative_mining_ratio=3, return_code=True)
    assert rc == -2
    lab[1, 1] = -1                                                               # a clean batch clears the codes
    op.MultiBoxTarget(dev(anc), dev(lab), dev(pred), negative_mining_ratio=3)
    op.MultiBoxTarget_check(2, dev(anc).device)


def test_target_run_to_run_deterministic(gpu_device):
    anc = mc.r50_anchors(512, 512)
    lab, pred = mc.target_inputs(anc, batch=6, seed=77)
    a = host(op.MultiBoxTarget(dev(anc), dev(lab), dev(pred), negative_mining_ratio=3))
    for _ in range(3):
        b = host(op.MultiBoxTarget(dev(anc), dev(lab), dev(pred), negative_mining_ratio=3))
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


@pytest.mark.gpu
def test_target_into_caller_buffers_on_another_stream(gpu_device):
    """op.MultiBoxTarget(out=...) (round 4: the graph node keeps its three outputs and runs the operator on a second HIP
    stream beside the segmentation decoder): the same bits as the allocating call, from a side stream ordered by events;
    buffers of the wrong shape are refused"""
    anc = mc.r50_anchors(512, 512)
    lab, pred = mc.target_inputs(anc, batch=4, seed=5)
    a, l, p = dev(anc), dev(lab), dev(pred)
    ref = host(op.MultiBoxTarget(a, l, p, negative_mining_ratio=3))
    B, N = lab.shape[0], anc.shape[1]
    out = (torch.full((B, N * 5), 7.0, device="cuda"), torch.full((B, N * 5), 7.0, device="cuda"), torch.full((B, N), 7.0, device="cuda"))
    side, ready, done = torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event()
    ready.record(torch.cuda.current_stream())
    side.wait_event(ready)
    with torch.cuda.stream(side):
        got = op.MultiBoxTarget(a, l, p, negative_mining_ratio=3, out=out)
        done.record(side)
    torch.cuda.current_stream().wait_event(done)
    assert all(g.data_ptr() == o.data_ptr() for g, o in zip(got, out))
    for x, y in zip(host(got), ref):
        np.testing.assert_array_equal(x, y)
    with pytest.raises(DspnError, match="out buffers"):
        op.MultiBoxTarget(a, l, p, out=(out[0], out[1], torch.zeros(B, N + 1, device="cuda")))


# ---------------------------------------------------------------- detection
def run_detection(anc, prob, loc, **kw):
    got = op.MultiBoxDetection(dev(prob), dev(loc), dev(anc), **kw).cpu().numpy()
    exp = om.multibox_detection(prob, loc, anc, **kw)
    np.testing.assert_array_equal(got, exp)
    return got


def test_detection_self_generated_regression_fixture(gpu_device):
    g = np.load(os.path.join(GOLDEN, "multibox_small.npz"))
    got = op.MultiBoxDetection(dev(g["cls_prob"]), dev(g["loc_pred"]), dev(g["anchors"]),
                               nms_threshold=.45, nms_topk=20).cpu().numpy()
    np.testing.assert_array_equal(got, g["det"])


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("kw", [dict(), dict(nms_topk=5), dict(force_suppress=True, clip=False),
                                dict(nms_threshold=-1.0), dict(nms_threshold=1.0, nms_topk=3),
                                dict(threshold=0.3, nms_threshold=0.3), dict(threshold=2.0)])
def test_detection_small_random(gpu_device, seed, kw):
    anc = mc.small_anchors(6, 7)
    prob, loc = mc.detection_inputs(anc, batch=3, num_classes=4, seed=seed, peaky=False)
    run_detection(anc, prob, loc, **kw)


@pytest.mark.parametrize("peaky", [True, False])
def test_detection_dspnet_shape(gpu_device, peaky):
    """symbol/multitask_symbol_builder.py:536-538: nms_thresh .5, nms_topk 400; peaky=False is
    the random-weights regime where nearly all 6132 rows are valid (worst case for sort + NMS)"""
    anc = mc.r50_anchors(512, 512)
    prob, loc = mc.detection_inputs(anc, batch=4, seed=31, peaky=peaky)
    got = run_detection(anc, prob, loc, nms_threshold=.5, force_suppress=False, nms_topk=400)
    assert ((got[..., 0] >= 0).sum(axis=1) > 10).all()


@pytest.mark.parametrize("num_classes,absent", [(21, ()), (81, (3, 40)), (3, (1,))])
def test_detection_many_classes_segments(gpu_device, num_classes, absent):
    """the suppression matrix and the greedy scan run per class segment of the class-grouped rows: many classes (segments
    far from 64-row aligned, some shorter than one block), classes that never win (empty segments), nms_topk leaving stale
    rows whose class is the pre-sort row's"""
    anc = mc.r50_anchors(512, 512)
    prob, loc = mc.detection_inputs(anc, batch=3, num_classes=num_classes, seed=77 + num_classes, peaky=False)
    for c in absent:
        prob[:, c] = 0.0
    for kw in (dict(nms_threshold=.5, nms_topk=400), dict(nms_threshold=.3), dict(nms_threshold=.5, nms_topk=5000)):
        run_detection(anc, prob, loc, **kw)


def test_detection_cityscapes_shape_force_suppress(gpu_device):
    anc = mc.r50_anchors(512, 1024)
    prob, loc = mc.detection_inputs(anc, batch=2, seed=32, peaky=True)
    run_detection(anc, prob, loc, nms_threshold=.45, force_suppress=True, nms_topk=400)


def test_detection_equal_scores_keep_anchor_order(gpu_device):
    anc = mc.small_anchors(8, 8)
    A = anc.shape[1]
    prob = np.full((2, 3, A), 1 / 3, np.float32)
    loc = np.zeros((2, A * 5), np.float32)
    run_detection(anc, prob, loc, nms_threshold=0.7)


def test_detection_single_class_and_no_valid_rows(gpu_device):
    anc = mc.small_anchors()
    A = anc.shape[1]
    out = run_detection(anc, np.ones((2, 1, A), np.float32), np.zeros((2, A * 5), np.float32))
    assert (out == -1).all()
