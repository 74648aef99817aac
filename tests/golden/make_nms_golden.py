#!/opt/conda/bin/python3.9
"""Generates tests/golden/nms_pixel.npz from the REFERENCE's own pixel-coordinate NMS code, run in the build
container:

  * cpu_nms      -- cython/cpu_nms.pyx:17-68 compiled unmodified by oracle/build_ref_cpu_nms.sh into
                    oracle/_ref/cpu_nms.cpython-39-*.so (Anaconda python3.9, Cython 0.29.24, numpy 1.26.4);
  * nms          -- detect/nms.py:24-58.  The module cannot be imported (its lines 2-3 import the compiled
                    cpu_nms / gpu_nms extension modules, the latter needs nvcc), so the one FunctionDef `nms` is
                    compiled from the file where it lies (ast) and executed against numpy; nothing else of the file
                    runs and no text of it is stored.

Run with:  /opt/conda/bin/python3.9 tests/golden/make_nms_golden.py      (after oracle/build_ref_cpu_nms.sh)

`np.int` -- the plain alias of the builtin `int` that numpy < 1.24 exported and cpu_nms.pyx uses as a dtype at run
time -- is restored before the call; nothing else about numpy is touched.

The file holds inputs and expected outputs only: dets_<k> (n,5) float32 [x1,y1,x2,y2,score], thresh_<k>, and the keep
lists keep_cpu_<k>, keep_py_<k> (int64; keep_cpu = [-2] where the reference raised ZeroDivisionError).  Scores inside one case are DISTINCT except in the cases named tie_*: the
reference orders by numpy's default (unstable) argsort()[::-1], so the order among equal scores is whatever the
numpy build does; order_<k> records the permutation numpy 1.26.4 produced here (the same expression on the same array),
and the tie cases are checked with that permutation handed to the oracle (tests/test_nms_pixel.py)."""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REF", "/root/reference")

np.int = int                                        # numpy < 1.24: `np.int is int`
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
import cpu_nms as ref_cpu                           # noqa: E402  (the compiled reference)


def load_py_nms():
    src = open(os.path.join(REF, "detect", "nms.py")).read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "nms"]
    assert len(fns) == 1
    mod = ast.Module(body=fns, type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, os.path.join(REF, "detect", "nms.py"), "exec"), ns)
    return ns["nms"]


def boxes(rng, n, extent, wmax, distinct=True, integer=False):
    x1 = rng.uniform(0, extent, n); y1 = rng.uniform(0, extent, n)
    w = rng.uniform(2, wmax, n); h = rng.uniform(2, wmax, n)
    d = np.stack([x1, y1, x1 + w, y1 + h, rng.uniform(0.01, 1, n)], 1).astype(np.float32)
    if integer:
        d[:, :4] = np.round(d[:, :4])
    if distinct:                                    # make float32 scores pairwise distinct
        s = np.sort(rng.permutation(np.arange(1, n + 1)).astype(np.float32))
        d[:, 4] = (rng.permutation(s) / np.float32(n + 1)).astype(np.float32)
        assert len(np.unique(d[:, 4])) == n
    return d


def main():
    py_nms = load_py_nms()
    rng = np.random.RandomState(20240233)
    cases = {}
    k = 0

    def add(name, dets, thresh):
        nonlocal k
        dets = np.ascontiguousarray(dets, np.float32)
        cases["name_%d" % k] = np.array(name)
        cases["dets_%d" % k] = dets
        cases["thresh_%d" % k] = np.float64(thresh)
        # the permutation both reference functions start from (for the tie_* cases: this numpy's order among equals)
        cases["order_%d" % k] = dets[:, 4].argsort()[::-1].astype(np.int64)
        try:
            cases["keep_cpu_%d" % k] = np.asarray(ref_cpu.cpu_nms(dets.copy(), float(thresh)), np.int64)
        except ZeroDivisionError:       # Cython's checked float division: a pair with union == 0 aborts the call
            cases["keep_cpu_%d" % k] = np.asarray([-2], np.int64)
        cases["keep_py_%d" % k] = np.asarray(py_nms(dets.copy(), thresh), np.int64)
        k += 1

    for n, extent, wmax in ((1, 50, 30), (2, 20, 30), (17, 60, 40), (100, 200, 80), (300, 300, 120), (1000, 400, 150)):
        for t in (0.3, 0.5, 0.95):
            add("random_n%d" % n, boxes(rng, n, extent, wmax), t)
    # dense cluster: long suppression chains
    add("cluster", boxes(rng, 400, 30, 60), 0.5)
    add("cluster_int", boxes(rng, 400, 40, 60, integer=True), 0.3)
    # overlaps exactly AT the threshold: A = 10x10 (area 100), B = 5x10 inside it -> IoU 50/100 = 0.5 exactly;
    # cpu_nms suppresses at ovr >= thresh, the numpy `nms` keeps ovr <= thresh
    eq = np.array([[0, 0, 9, 9, 0.9], [0, 0, 4, 9, 0.8], [20, 20, 29, 29, 0.7], [20, 20, 29, 24, 0.6],
                   [0, 0, 9, 4, 0.5]], np.float32)
    add("threshold_equal", eq, 0.5)
    add("threshold_equal_lo", eq, 0.25)
    # degenerate boxes (x2 < x1: zero or negative "area" with the +1 convention)
    deg = boxes(rng, 50, 60, 30)
    deg[::7, 2] = deg[::7, 0] - 1.5                 # width -0.5: negative area, unions stay non-zero
    deg[3::11, 3] = deg[3::11, 1] - 3.0
    add("degenerate", deg, 0.5)
    # two zero-area boxes: union == 0.  cpu_nms.pyx raises ZeroDivisionError (recorded as keep_cpu = [-2]); the numpy
    # `nms` gets nan, and `nan <= thresh` being false drops the second zero-area box
    zero = np.array([[5, 5, 4, 9, 0.9], [30, 30, 29, 40, 0.8], [0, 0, 9, 9, 0.7]], np.float32)
    add("degenerate_zero_union", zero, 0.5)
    # tied scores (order among equals = numpy's unstable sort; see the module docstring)
    tie = boxes(rng, 64, 100, 40, distinct=False)
    tie[:, 4] = np.float32(0.5)
    add("tie_all_equal", tie, 0.5)
    tie2 = boxes(rng, 200, 150, 60, distinct=False)
    tie2[:, 4] = (np.round(tie2[:, 4] * 8) / 8).astype(np.float32)
    add("tie_quantised", tie2, 0.5)
    cases["count"] = np.int64(k)
    cases["numpy_version"] = np.array(np.__version__)
    out = os.path.join(HERE, "nms_pixel.npz")
    np.savez_compressed(out, **cases)
    print("wrote", out, k, "cases,", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
