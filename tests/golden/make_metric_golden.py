#!/usr/bin/env python3
"""Generates tests/golden/metric_pure.npz from the REFERENCE's own code, run in the build container.

The reference's metric / initialisation modules import mxnet at the top and cannot be imported here, but four of their
functions are plain numpy and do not touch `self` or mxnet:
    evaluate/eval_metric.py  MApMetric._recall_prec (:195-206), MApMetric._average_precision (:208-235),
                             VOC07MApMetric._average_precision (:254-276)
    multi_init.py            upsample_filt (:13-21)
Each FunctionDef is compiled from the file where it lies (ast) and executed against numpy; nothing else of the files
runs and no text of them is stored.  The file holds inputs and the outputs the reference code produced:
    rp_<k>: records (n,2) [score, flag 0 ignore / 1 tp / 2 fp], count -> recall, prec
    ap_<k> / ap07_<k>: the two average-precision integrals of that (recall, prec)
    filt_<s>: upsample_filt(s), s = 1..8
Run with:  python3 tests/golden/make_metric_golden.py"""
import ast
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("REF", "/root/reference")


def extract(path, func, cls=None):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    body = tree.body
    if cls is not None:
        body = [n for n in body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    fn = [n for n in body if isinstance(n, ast.FunctionDef) and n.name == func]
    assert len(fn) == 1, (path, cls, func)
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), os.path.join(REF, path), "exec"), ns)
    return ns[func]


def main():
    recall_prec = extract("evaluate/eval_metric.py", "_recall_prec", "MApMetric")
    ap = extract("evaluate/eval_metric.py", "_average_precision", "MApMetric")
    ap07 = extract("evaluate/eval_metric.py", "_average_precision", "VOC07MApMetric")
    filt = extract("multi_init.py", "upsample_filt")
    rng = np.random.RandomState(7)
    out = {}
    cases = []
    for n, count in ((1, 1), (5, 3), (40, 12), (200, 50), (200, 0), (30, 5), (17, 40)):
        rec = np.stack([rng.rand(n), rng.randint(0, 3, n).astype(float)], 1)
        cases.append((rec, count))
    cases.append((np.array([[0.9, 1.], [0.9, 2.], [0.5, 1.], [0.5, 1.], [0.1, 0.]]), 4))      # tied scores
    cases.append((np.array([[0.7, 2.], [0.6, 2.]]), 3))                                          # no true positive
    cases.append((np.array([[0.7, 0.], [0.6, 0.]]), 2))                                          # everything ignored
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k, (rec, count) in enumerate(cases):
            r, p = recall_prec(None, rec.copy(), count)
            out["rp_%d_records" % k] = rec
            out["rp_%d_count" % k] = np.int64(count)
            out["rp_%d_recall" % k] = np.asarray(r, np.float64)
            out["rp_%d_prec" % k] = np.asarray(p, np.float64)
            out["ap_%d" % k] = np.float64(ap(None, r, p))
            out["ap07_%d" % k] = np.float64(ap07(None, r, p))
    out["cases"] = np.int64(len(cases))
    for s in range(1, 9):
        out["filt_%d" % s] = np.asarray(filt(s), np.float64)
    path = os.path.join(HERE, "metric_pure.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(cases), "cases,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
