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
