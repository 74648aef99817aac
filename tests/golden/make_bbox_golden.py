#!/opt/conda/bin/python3.9
"""Generates tests/golden/bbox_overlaps.npz from the REFERENCE's own box-overlap function, run in the build container:
cython/bbox.pyx:15-55 (`bbox_overlaps_cython`, float64, the "+1" pixel convention) compiled unmodified by
oracle/build_ref_cpu_nms.sh into oracle/_ref/bbox.cpython-39-*.so (Anaconda python3.9, Cython 0.29.24, numpy 1.26.4).

Run with:  /opt/conda/bin/python3.9 tests/golden/make_bbox_golden.py      (after oracle/build_ref_cpu_nms.sh)

`np.float` -- the plain alias of the builtin `float` that numpy < 1.24 exported and bbox.pyx assigns to its module-level
DTYPE at import time -- is restored before the import; nothing else about numpy is touched.

The file holds inputs and expected outputs only: boxes_<k> (N,4) float64, query_<k> (K,4) float64, overlaps_<k> (N,K)
float64 as the reference returned them."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

np.float = float                                    # numpy < 1.24: `np.float is float`
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
import bbox as ref_bbox                             # noqa: E402  (the compiled reference)


def boxes(rng, n, extent, wmax, integer=False):
    x1 = rng.uniform(0, extent, n); y1 = rng.uniform(0, extent, n)
    w = rng.uniform(0, wmax, n); h = rng.uniform(0, wmax, n)
    d = np.stack([x1, y1, x1 + w, y1 + h], 1)
    return np.round(d) if integer else d


def main():
    rng = np.random.RandomState(20241002)
    cases = {}
    k = 0

    def add(name, b, q):
        nonlocal k
        b = np.ascontiguousarray(b, np.float64).reshape(-1, 4)
        q = np.ascontiguousarray(q, np.float64).reshape(-1, 4)
        cases["name_%d" % k] = np.array(name)
        cases["boxes_%d" % k] = b
        cases["query_%d" % k] = q
        cases["overlaps_%d" % k] = np.asarray(ref_bbox.bbox_overlaps_cython(b, q), np.float64)
        k += 1

    for n, kq, extent, wmax in ((1, 1, 20, 30), (7, 3, 40, 30), (100, 37, 200, 80), (513, 129, 600, 150), (2000, 64, 1000, 300)):
        add("random_%dx%d" % (n, kq), boxes(rng, n, extent, wmax), boxes(rng, kq, extent, wmax))
    add("integer", boxes(rng, 300, 100, 40, integer=True), boxes(rng, 50, 100, 40, integer=True))
    # boxes that touch: with the +1 convention a shared edge column still intersects (iw = 1); one pixel apart iw = 0,
    # which is NOT > 0 and stays exactly 0
    touch_b = np.array([[0, 0, 9, 9], [10, 0, 19, 9], [11, 0, 20, 9], [0, 10, 9, 19], [0, 11, 9, 20]], np.float64)
    add("touching", touch_b, np.array([[0, 0, 9, 9], [9, 9, 12, 12]], np.float64))
    same = boxes(rng, 40, 80, 50)
    add("identical", same, same.copy())
    # degenerate boxes (x2 < x1): negative widths, areas of either sign; unions of zero never reach the division because
    # iw / ih <= 0 short-circuits first in every such pair below
    deg = boxes(rng, 60, 60, 30)
    deg[::5, 2] = deg[::5, 0] - 2.5
    deg[3::7, 3] = deg[3::7, 1] - 4.0
    add("degenerate", deg, boxes(rng, 20, 60, 30))
    add("large_coordinates", boxes(rng, 64, 1e6, 5e4) + 3e6, boxes(rng, 16, 1e6, 5e4) + 3e6)
    add("empty_boxes", np.zeros((0, 4)), boxes(rng, 5, 10, 10))
    add("empty_query", boxes(rng, 5, 10, 10), np.zeros((0, 4)))
    cases["count"] = np.int64(k)
    cases["numpy_version"] = np.array(np.__version__)
    out = os.path.join(HERE, "bbox_overlaps.npz")
    np.savez_compressed(out, **cases)
    print("wrote", out, k, "cases,", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
