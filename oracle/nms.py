"""CPU restatement of the reference's pixel-coordinate NMS.  TEST INFRASTRUCTURE ONLY.

PARITY STATUS: PINNED against the reference's own code run in the build container.  cython/cpu_nms.pyx:17-68 is
compiled unmodified (oracle/build_ref_cpu_nms.sh -> oracle/_ref/, Anaconda python3.9 + Cython 0.29 + numpy 1.26 of
this image) and the `nms` function of detect/nms.py:24-58 is executed from the file where it lies
(tests/golden/make_nms_golden.py); their keep lists on 27 cases -- n = 1..1000, thresholds 0.3 / 0.5 / 0.95, dense
clusters, overlaps exactly at the threshold, degenerate boxes, tied scores -- are committed as
tests/golden/nms_pixel.npz and this file reproduces all of them (tests/test_nms_pixel.py).
Both functions are restated in float32, operation for operation.  Ties in score: the reference uses numpy's unstable
argsort()[::-1], whose order among equal scores depends on the numpy build; a stable ascending sort reversed is used
here (higher index first), and `order=` takes an explicit permutation (the golden file records the one numpy 1.26.4
produced).  One behaviour is NOT reproduced: cpu_nms.pyx raises ZeroDivisionError for a pair whose union is exactly 0
(Cython's checked division); here, as in the numpy `nms`, the quotient is nan and the pair is not suppressed."""
import numpy as np


def _order(scores):
    return np.argsort(scores, kind="stable")[::-1]


def nms(dets, thresh, order=None):
    """detect/nms.py:24-58"""
    dets = np.asarray(dets, np.float32)
    x1, y1, x2, y2, scores = (dets[:, i] for i in range(5))
    one = np.float32(1)
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = _order(scores) if order is None else np.asarray(order)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        xx1 = np.maximum(x1[i], x1[order[1:]]); yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]]); yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(np.float32(0), xx2 - xx1 + one)
        h = np.maximum(np.float32(0), yy2 - yy1 + one)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[np.where(ovr <= np.float32(thresh))[0] + 1]
    return keep


def cpu_nms(dets, thresh, order=None):
    """cython/cpu_nms.pyx:17-68 (suppress when ovr >= thresh)"""
    dets = np.asarray(dets, np.float32)
    x1, y1, x2, y2, scores = (dets[:, i] for i in range(5))
    one = np.float32(1)
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = _order(scores) if order is None else np.asarray(order)
    n = dets.shape[0]
    suppressed = np.zeros(n, bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(int(i))
        rest = order[_i + 1:]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + one)
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + one)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr >= np.float32(thresh)]] = True
    return keep


def bbox_overlaps(boxes, query_boxes):
    """cython/bbox.pyx:15-55 (`bbox_overlaps_cython`): (N, 4) x (K, 4) float64 -> (N, K) float64 overlaps in the "+1"
    pixel convention, operation for operation in float64 (iw first; a pair with iw <= 0 or ih <= 0 stays exactly 0 and
    never reaches the division).  PINNED: the reference file compiles unmodified (oracle/build_ref_cpu_nms.sh) and its
    outputs on 12 cases are committed as tests/golden/bbox_overlaps.npz, reproduced bit for bit (tests/test_nms_pixel.py)."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    q = np.asarray(query_boxes, np.float64).reshape(-1, 4)
    box_area = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)                       # (K,)
    iw = np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0]) + 1
    ih = np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1]) + 1
    ua = (b[:, 2] - b[:, 0] + 1)[:, None] * (b[:, 3] - b[:, 1] + 1)[:, None] + box_area[None, :] - iw * ih
    with np.errstate(divide="ignore", invalid="ignore"):
        ov = iw * ih / ua
    return np.where((iw > 0) & (ih > 0), ov, 0.0)
