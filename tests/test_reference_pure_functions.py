"""Pinned against the reference's own code (tests/golden/make_metric_golden.py executes the four plain-numpy functions
of evaluate/eval_metric.py and multi_init.py from the files where they lie): recall / precision, the two
average-precision integrals and the bilinear deconvolution kernel -- product code AND oracle restatement."""
import os
import warnings

import numpy as np

from dspnet_amd import multi_init
from dspnet_amd.evaluate.eval_metric import MApMetric, VOC07MApMetric
from oracle import metrics as om

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metric_pure.npz"))


def _same(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=0, atol=0)


def test_recall_precision_and_average_precision_match_the_reference():
    m, m07 = MApMetric(class_names=["a"]), VOC07MApMetric(class_names=["a"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(int(Z["cases"])):
            rec, count = Z["rp_%d_records" % k], int(Z["rp_%d_count" % k])
            r, p = m._recall_prec(rec.copy(), count)
            _same(r, Z["rp_%d_recall" % k]); _same(p, Z["rp_%d_prec" % k])
            _same(m._average_precision(r, p), Z["ap_%d" % k])
            _same(m07._average_precision(r, p), Z["ap07_%d" % k])
            # the oracle restatement (oracle/metrics.py: MApOracle.get() computes recall / precision inline)
            _same(om.MApOracle._ap(Z["rp_%d_recall" % k], Z["rp_%d_prec" % k]), Z["ap_%d" % k])
            _same(om.MApOracle._ap07(Z["rp_%d_recall" % k], Z["rp_%d_prec" % k]), Z["ap07_%d" % k])
            for voc07, key in ((False, "ap_%d"), (True, "ap07_%d")):
                o = om.MApOracle(voc07=voc07)
                o.records[0], o.counts[0] = rec.copy(), count
                _same(o.get()[0], Z[key % k])


def test_upsample_filt_matches_the_reference():
    for s in range(1, 9):
        _same(multi_init.upsample_filt(s), Z["filt_%d" % s])
