"""Pixel-coordinate NMS: oracle self-checks on the CPU, HIP vs oracle on the GPU."""
import numpy as np
import pytest

from oracle import nms as onms


def boxes(n, seed, size=640, ties=False):
    g = np.random.Generator(np.random.PCG64(seed))
    x1 = g.integers(0, size - 40, n).astype(np.float32); y1 = g.integers(0, size - 40, n).astype(np.float32)
    w = g.integers(5, 200, n).astype(np.float32); h = g.integers(5, 200, n).astype(np.float32)
    s = g.permutation(n).astype(np.float32) / n if not ties else np.round(g.random(n) * 8).astype(np.float32) / 8
    return np.stack([x1, y1, np.minimum(x1 + w, size - 1), np.minimum(y1 + h, size - 1), s], 1).astype(np.float32)


def test_oracle_hand_case():
    """two heavily overlapping boxes and a distant one: the lower-scored overlapping box goes"""
    d = np.array([[10, 10, 50, 50, .9], [12, 12, 52, 52, .8], [200, 200, 240, 240, .7]], np.float32)
    assert onms.nms(d, 0.5) == [0, 2] and onms.cpu_nms(d, 0.5) == [0, 2]
    # IoU of boxes 0 and 1 with the +1 convention: inter 39*39, union 2*41*41 - 39*39
    iou = np.float32(39 * 39) / np.float32(2 * 41 * 41 - 39 * 39)
    assert onms.nms(d, float(iou)) == [0, 1, 2]         # "<= thresh" keeps an overlap equal to the threshold
    assert onms.cpu_nms(d, float(iou)) == [0, 2]        # ">= thresh" suppresses it


def test_oracle_variants_agree_off_threshold():
    d = boxes(300, 3)
    assert onms.nms(d, 0.45) == onms.cpu_nms(d, 0.45)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (63, 2), (64, 3), (65, 4), (500, 5), (2000, 6), (8192, 7)])
@pytest.mark.parametrize("thresh", [0.3, 0.5, 0.95])
def test_hip_nms_matches_oracle(gpu_device, n, seed, thresh):
    from dspnet_amd.detect import nms as dn
    d = boxes(n, seed)
    assert dn.nms(d, thresh) == onms.nms(d, thresh)
    assert dn.gpu_nms(d, thresh, 0) == onms.nms(d, thresh)
    assert dn.cpu_nms(d, thresh) == onms.cpu_nms(d, thresh)
    assert dn.py_nms_wrapper(thresh)(d) == onms.nms(d, thresh)


@pytest.mark.gpu
def test_hip_nms_ties_and_threshold_equality(gpu_device):
    from dspnet_amd.detect import nms as dn
    d = boxes(400, 11, ties=True)                       # many equal scores: order = higher index first
    assert dn.nms(d, 0.5) == onms.nms(d, 0.5)
    e = np.array([[10, 10, 50, 50, .9], [12, 12, 52, 52, .8], [200, 200, 240, 240, .7]], np.float32)
    iou = float(np.float32(39 * 39) / np.float32(2 * 41 * 41 - 39 * 39))
    assert dn.nms(e, iou) == [0, 1, 2] and dn.cpu_nms(e, iou) == [0, 2]
    assert dn.nms(np.zeros((0, 5), np.float32), 0.5) == []


# ---------------------------------------------------------------------------------------------------------------
# Golden vectors produced by the REFERENCE's own code in the build container (tests/golden/make_nms_golden.py):
# cython/cpu_nms.pyx compiled unmodified (oracle/build_ref_cpu_nms.sh) and the `nms` function of detect/nms.py.
def _golden():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nms_pixel.npz"))
    for k in range(int(z["count"])):
        yield (str(z["name_%d" % k]), z["dets_%d" % k], float(z["thresh_%d" % k]), z["order_%d" % k],
               z["keep_cpu_%d" % k].tolist(), z["keep_py_%d" % k].tolist())


def test_oracle_reproduces_reference_golden_vectors():
    """pins oracle/nms.py: every keep list of the reference-generated fixture, indices AND order"""
    import warnings
    seen = set()
    for name, d, t, order, keep_cpu, keep_py in _golden():
        seen.add(name)
        tie = name.startswith("tie_")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            # tied scores: the reference's order among equals is numpy's unstable sort; hand its permutation over
            got_py = onms.nms(d, t, order=order if tie else None)
            got_cpu = onms.cpu_nms(d, t, order=order if tie else None)
        assert got_py == keep_py, (name, t)
        if keep_cpu == [-2]:        # cpu_nms.pyx raised ZeroDivisionError (union == 0); documented difference
            assert name == "degenerate_zero_union"
        else:
            assert got_cpu == keep_cpu, (name, t)
        if not tie:                 # distinct scores: the oracle's own stable order is the reference's order
            assert np.array_equal(onms._order(d[:, 4]), order), name
    assert {"random_n1000", "cluster", "threshold_equal", "degenerate", "tie_all_equal", "tie_quantised"} <= seen


def test_golden_threshold_equal_case_separates_the_two_variants():
    case = {n: (d, t, kc, kp) for n, d, t, _, kc, kp in _golden()}["threshold_equal"]
    d, t, keep_cpu, keep_py = case
    assert t == 0.5 and keep_py == [0, 1, 2, 3, 4] and keep_cpu == [0, 2]   # IoU == 0.5 exactly: "<=" keeps, ">=" drops


@pytest.mark.gpu
def test_hip_nms_reproduces_reference_golden_vectors(gpu_device):
    """the HIP kernel against the reference-generated keep lists directly (distinct-score cases; with tied scores the
    reference's order is numpy-build dependent and the kernel follows the oracle's documented order instead)"""
    from dspnet_amd.detect import nms as dn
    n = 0
    for name, d, t, order, keep_cpu, keep_py in _golden():
        if name.startswith("tie_"):
            assert dn.nms(d, t) == onms.nms(d, t) and dn.cpu_nms(d, t) == onms.cpu_nms(d, t)
            continue
        assert dn.nms(d, t) == keep_py, (name, t)
        if name == "degenerate_zero_union":          # 0/0 overlap: `ovr > thresh` (nms_kernel.cu:68) keeps what numpy's `ovr <= thresh` drops
            assert keep_py == [0, 2] and dn.gpu_nms(d, t, 0) == [0, 1, 2]
        else:
            assert dn.gpu_nms(d, t, 0) == keep_py, (name, t)
        if keep_cpu != [-2]:
            assert dn.cpu_nms(d, t) == keep_cpu, (name, t)
        n += 1
    assert n >= 20


# ---------------------------------------------------------------------------------------------------------------
# bbox_overlaps (cython/bbox.pyx:15-55), pinned the same way: the .pyx compiles unmodified (oracle/build_ref_cpu_nms.sh),
# tests/golden/make_bbox_golden.py stores its inputs and outputs.
def _bbox_golden():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bbox_overlaps.npz"))
    for k in range(int(z["count"])):
        yield str(z["name_%d" % k]), z["boxes_%d" % k], z["query_%d" % k], z["overlaps_%d" % k]


def test_bbox_overlaps_oracle_reproduces_reference_golden_vectors():
    """pins oracle/nms.py: bbox_overlaps bit for bit (float64) on every case, incl. empty inputs and touching boxes"""
    seen = set()
    for name, b, q, exp in _bbox_golden():
        seen.add(name)
        got = onms.bbox_overlaps(b, q)
        assert got.shape == exp.shape and got.dtype == np.float64
        assert np.array_equal(got, exp), name
    assert {"random_2000x64", "touching", "identical", "degenerate", "empty_boxes", "empty_query"} <= seen
    # hand case: a shared edge column still intersects with the +1 convention (iw = 1); one pixel apart does not
    name, b, q, exp = [c for c in _bbox_golden() if c[0] == "touching"][0]
    assert exp[0, 0] == 1.0 and exp[1, 0] == 0.0 and exp[2, 0] == 0.0
    assert exp[0, 1] == 1.0 / (100 + 16 - 1)            # boxes share exactly the pixel (9, 9)


@pytest.mark.gpu
def test_hip_bbox_overlaps_reproduces_reference_golden_vectors(gpu_device):
    """dspn_bbox_overlaps_f64 against the reference-generated matrices: identical bits"""
    import torch
    from dspnet_amd.detect import nms as dn
    for name, b, q, exp in _bbox_golden():
        got = dn.bbox_overlaps(b, q)
        assert isinstance(got, np.ndarray) and got.shape == exp.shape and got.dtype == np.float64
        assert np.array_equal(got, exp), name
    g = np.random.Generator(np.random.PCG64(5))
    b = g.uniform(0, 500, (5000, 4)); b[:, 2:] += b[:, :2]
    q = g.uniform(0, 500, (200, 4)); q[:, 2:] += q[:, :2]
    assert np.array_equal(dn.bbox_overlaps_cython(b, q), onms.bbox_overlaps(b, q))         # several query blocks
    t = dn.bbox_overlaps(torch.from_numpy(b).cuda(), torch.from_numpy(q).cuda())
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), onms.bbox_overlaps(b, q))
