"""Pinned against the reference's own code (tests/golden/make_config_golden.py): `get_config` of
symbol/multitask_symbol_factory.py for every network x data_shape it defines, and the segmentation look-up table that
dataset/iterator.py:358-363 builds from dataset/cs_labels.py."""
import json
import os

import numpy as np
import pytest

from dspnet_amd.dataset.iterator import seg_lut
from dspnet_amd.symbol.multitask_symbol_factory import get_config

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "factory_config.json")))


def _norm(v):
    if isinstance(v, (list, tuple)):
        return [_norm(x) for x in v]
    if isinstance(v, float):
        return round(v, 12)
    return v


@pytest.mark.parametrize("case", G["get_config"], ids=lambda c: "%s-%d" % (c["network"], c["data_shape"]))
def test_get_config_matches_the_reference(case):
    if "error" in case:
        with pytest.raises(NotImplementedError):
            get_config(case["network"], case["data_shape"])
        return
    ours = dict(get_config(case["network"], case["data_shape"]))
    ours.pop("kwargs", None)
    ref = case["config"]
    # every key the reference returns, value for value (network name, layer names, filters, strides, pads, sizes, ratios,
    # normalizations, steps, num_layers / image_shape of the resnet preset)
    for k, v in ref.items():
        if k == "data_shape":
            continue      # the reference returns its argument; this build normalises (C, H, W) tuples to H first
        assert k in ours, k
        assert _norm(ours[k]) == _norm(v), (k, ours[k], v)


def test_segmentation_lut_matches_the_reference():
    assert np.array_equal(seg_lut(), np.asarray(G["seg_lut"], np.uint8))
    ids = sorted(i for _, i, t in G["labels"] if t >= 0)
    assert ids == list(range(35))       # what seg_lut() hard-codes: ids 0 .. 34 keep their value, everything else is 255
