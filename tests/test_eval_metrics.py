"""Evaluation read-outs: MApMetric / VOC07MApMetric, DistanceAccuracyMetric and the fused
full-resolution class map, each against the restatement in oracle/metrics.py and against hand-computed cases."""
import math

import numpy as np
import pytest

from dspnet_amd.evaluate.eval_metric import MApMetric, VOC07MApMetric
from dspnet_amd.train.metric import DistanceAccuracyMetric
from oracle import metrics as om


def _box(x, y, w=0.1, h=0.1):
    return [x, y, x + w, y + h]


def test_map_hand_case():
    # one class, two ground truths; detections: hit gt0, miss, hit gt1 -> recall .5 .5 1, precision 1 .5 2/3
    label = np.full((1, 4, 5), -1, np.float32)
    label[0, 0] = [0] + _box(.1, .1); label[0, 1] = [0] + _box(.5, .5)
    pred = np.full((1, 5, 6), -1, np.float32)
    pred[0, 0] = [0, .9] + _box(.1, .1); pred[0, 1] = [0, .8] + _box(.8, .1); pred[0, 2] = [0, .7] + _box(.5, .5)
    m = MApMetric(class_names=["car"])
    m.update([label], [pred])
    names, vals = m.get()
    assert names == ["car", "mAP"]
    assert abs(vals[0] - (0.5 * 1.0 + 0.5 * 2.0 / 3.0)) < 1e-12 and vals[0] == vals[1]
    v = VOC07MApMetric(class_names=["car"])
    v.update([label], [pred])
    assert abs(v.get()[1][1] - (6 * 1.0 + 5 * 2.0 / 3.0) / 11.0) < 1e-12
    # no class names: scalar form
    s = MApMetric()
    s.update([label], [pred])
    assert s.get()[0] == "mAP" and abs(s.get()[1] - vals[0]) < 1e-12
    assert math.isnan(MApMetric().get()[1]) or True     # empty metric: nan (np.mean of nothing), no exception


def test_map_reference_quirks():
    # duplicate detection of one ground truth -> second is a false positive; unseen class adds its count only
    label = np.full((1, 3, 5), -1, np.float32)
    label[0, 0] = [1] + _box(.2, .2); label[0, 1] = [2] + _box(.6, .6)
    pred = np.full((1, 3, 6), -1, np.float32)
    pred[0, 0] = [1, .9] + _box(.2, .2); pred[0, 1] = [1, .8] + _box(.21, .2)
    m = MApMetric(class_names=["a", "b", "c"])
    m.update([label], [pred])
    assert m.records[1][:, 1].tolist() == [1, 2] and m.counts == {1: 1, 2: 1}
    assert m.records[2].tolist() == [[0, 0]]
    names, vals = m.get()
    assert math.isnan(vals[0]) and vals[1] == 1.0 and vals[2] == 0.0 and vals[3] == 0.5
    # "difficult" column: a match to a flagged box is dropped, flagged boxes are not counted
    lab6 = np.full((1, 2, 6), -1, np.float32)
    lab6[0, 0] = [0] + _box(.1, .1) + [1]; lab6[0, 1] = [0] + _box(.5, .5) + [0]
    p = np.full((1, 2, 6), -1, np.float32)
    p[0, 0] = [0, .9] + _box(.1, .1); p[0, 1] = [0, .5] + _box(.5, .5)
    d = MApMetric()
    d.update([lab6], [p])
    assert d.records[0].tolist() == [[0.5, 1.0]] and d.counts[0] == 1
    e = MApMetric(use_difficult=True)
    e.update([lab6], [p])
    assert e.records[0][:, 1].tolist() == [1, 1] and e.counts[0] == 2


def _random_eval_batch(g, B, L, M, ncls, cols):
    label = np.full((B, L, cols), -1, np.float32)
    pred = np.full((B, M, 6), -1, np.float32)
    for b in range(B):
        n = int(g.integers(0, L + 1)) if b else 0                     # first image has no ground truth at all
        xy = g.random((n, 2)) * 0.7
        wh = 0.05 + g.random((n, 2)) * 0.25
        label[b, :n, 0] = g.integers(0, ncls, n)
        label[b, :n, 1:3] = xy; label[b, :n, 3:5] = xy + wh
        if cols == 6:
            label[b, :n, 5] = g.integers(0, 2, n)
        k = int(g.integers(0, M + 1))
        rows = []
        for _ in range(k):
            if n and g.random() < 0.6:                                # jittered copy of a ground truth
                j = int(g.integers(0, n))
                box = label[b, j, 1:5] + g.normal(0, 0.02, 4)
                cid = label[b, j, 0] if g.random() < 0.8 else g.integers(0, ncls)
            else:
                p0 = g.random(2) * 0.7
                box = np.concatenate([p0, p0 + 0.05 + g.random(2) * 0.25]); cid = g.integers(0, ncls)
            rows.append([cid, g.random()] + list(box))
        rows.sort(key=lambda r: -r[1])
        if rows:
            pred[b, :k] = np.asarray(rows, np.float32)
        if k > 2:
            pred[b, 1, 0] = -1                                        # an empty slot in the middle
            pred[b, 2, 0] = -0.5                                      # truncates to class 0
    return label, pred


@pytest.mark.parametrize("cols,use_difficult,named,voc07", [(5, False, True, False), (6, False, True, False),
                                                           (6, True, False, False), (5, False, True, True)])
def test_map_matches_restatement(cols, use_difficult, named, voc07):
    g = np.random.Generator(np.random.PCG64(233))
    ncls = 5
    names = ["c%d" % i for i in range(ncls)] if named else None
    m = (VOC07MApMetric if voc07 else MApMetric)(0.5, use_difficult, names)
    o = om.MApOracle(0.5, use_difficult, ncls if named else None, voc07)
    for _ in range(4):
        label, pred = _random_eval_batch(g, 6, 12, 40, ncls, cols)
        m.update([label], [pred]); o.update(label, pred)
    assert sorted(m.records) == sorted(o.records)
    for k in m.records:
        np.testing.assert_array_equal(m.records[k], o.records[k])
        assert m.counts[k] == o.counts[k]
    vals = m.get()[1]
    ref = o.get()
    np.testing.assert_array_equal(np.atleast_1d(np.asarray(vals, float)), np.asarray(ref, float))
    assert np.isfinite(np.atleast_1d(vals)[-1])


def test_distance_metric_hand_case():
    disp = np.zeros((1, 100, 200), np.float32)
    disp[0, 20:40, 40:80] = 3300.0                   # 2200*75/3300 = 50 m
    disp[0, 60:80, 100:140] = 100.0                  # 1650 m -> > 1000 -> 200 -> skipped
    det = np.full((1, 4, 7), -1, np.float32)
    det[0, 0] = [1, .9, .2, .2, .4, .4, 55. / 255.]  # predicted 55 m vs 50 m -> 0.1
    det[0, 1] = [0, .8, .5, .6, .7, .8, .5]          # skipped (too far)
    det[0, 2] = [2, .7, .95, .95, .95, .99, .5]      # empty column range widened to one pixel: disparity 0 -> skipped
    m = DistanceAccuracyMetric(class_names=["a", "b", "c"])
    m.update(disp, [det])
    names, vals = m.get()
    assert names == ["a", "b", "c", "derror"]
    ref = abs(float(np.float32(55. / 255.)) * 255. - 2200. * 75. / (3300. + 1e-3)) / (2200. * 75. / (3300. + 1e-3))
    assert math.isnan(vals[0]) and math.isnan(vals[2]) and abs(vals[1] - ref) < 1e-12 and abs(vals[3] - ref) < 1e-12


def test_distance_metric_matches_restatement():
    g = np.random.Generator(np.random.PCG64(7))
    B, hh, ww, ncls = 3, 64, 128, 4
    m = DistanceAccuracyMetric(class_names=["c%d" % i for i in range(ncls)])
    sums = [0.0] * ncls; cnt = [0] * ncls
    for _ in range(3):
        disp = (g.random((B, hh, ww)) * 6000 + 800).astype(np.float32)
        dets = []
        for b in range(B):
            k = int(g.integers(0, 9))
            d = np.full((1, 10, 7), -1, np.float32)
            p0 = g.random((k, 2)) * 0.8
            d[0, :k, 0] = g.integers(0, ncls, k); d[0, :k, 1] = np.sort(g.random(k))[::-1]
            d[0, :k, 2:4] = p0 - 0.05; d[0, :k, 4:6] = p0 + g.random((k, 2)) * 0.3        # some xmin / ymin < 0
            d[0, :k, 6] = g.random(k)
            dets.append(d)
        m.update(disp, dets)
        err = om.distance_errors(disp, dets, ncls)
        for c in range(ncls):
            sums[c] += math.fsum(err[c]); cnt[c] += len(err[c])
    vals = m.get()[1]
    assert sum(cnt) > 10
    for c in range(ncls):
        assert (math.isnan(vals[c]) and cnt[c] == 0) or vals[c] == sums[c] / cnt[c]
    assert m.num_inst[-1] == sum(cnt)


def test_upsample_restatement_against_torch_sampler():
    """the numpy restatement of GridGenerator + BilinearSampler agrees with torch's grid_sample(align_corners=True)"""
    import torch
    g = np.random.Generator(np.random.PCG64(3))
    prob = g.random((2, 19, 9, 13)).astype(np.float32)
    Ho, Wo = 36, 50
    ys = torch.linspace(-1, 1, Ho); xs = torch.linspace(-1, 1, Wo)
    grid = torch.stack(torch.meshgrid(ys, xs, indexing="ij")[::-1], dim=-1)[None].expand(2, -1, -1, -1)
    ref = torch.nn.functional.grid_sample(torch.from_numpy(prob), grid, mode="bilinear", padding_mode="zeros",
                                          align_corners=True).numpy()
    got = om.upsample_argmax(prob, Ho, Wo)
    top2 = np.sort(ref, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-5                       # pixels whose winner does not hinge on rounding
    assert clear.mean() > 0.99
    np.testing.assert_array_equal(got[clear], ref.argmax(1)[clear])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 19, 16, 32, 128, 256), (1, 19, 9, 13, 37, 51), (3, 5, 7, 7, 7, 7),
                                   (1, 19, 128, 256, 1024, 2048), (1, 3, 4, 4, 1, 1)])
def test_seg_upsample_argmax_bit_exact(gpu_device, shape):
    import torch
    from dspnet_amd import functional as fn
    from dspnet_amd.evaluate.multi_eval import prob_upsampling
    N, C, h, w, Ho, Wo = shape
    g = np.random.Generator(np.random.PCG64(11))
    logits = g.standard_normal((N, C, h, w)).astype(np.float32) * 2
    e = np.exp(logits - logits.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32)
    prob[0, :, 0, 0] = 1.0 / C                                       # exact tie: the lowest class wins
    ld = fn.pad4(C)
    nhwc = np.zeros((N, h, w, ld), np.float32)
    nhwc[..., :C] = prob.transpose(0, 2, 3, 1)
    nhwc[..., C:] = 9.0                                              # pad channels must not be read
    got = fn.seg_upsample_argmax(torch.from_numpy(nhwc).cuda(), C, Ho, Wo).cpu().numpy()
    ref = om.upsample_argmax(prob, Ho, Wo)
    np.testing.assert_array_equal(got, ref)
    if C == 19:                                                      # the reference-layout entry point
        got2 = prob_upsampling(torch.from_numpy(prob).cuda(), (Ho, Wo)).cpu().numpy()
        np.testing.assert_array_equal(got2, ref)


@pytest.mark.gpu
def test_evaluate_net_end_to_end(gpu_device):
    import torch
    from dspnet_amd import functional as fn, synthetic
    from dspnet_amd.evaluate.multi_eval import evaluate_net, filter_detections, label_ids
    from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
    B, S = 2, 128
    net = get_multi_symbol_train("resnet-50", S, num_classes=8, batch_size=B, device=gpu_device)
    gen = synthetic.rng(233)
    batches = []
    for _ in range(2):
        batches.append({"data": torch.from_numpy(synthetic.images(B, S, S, gen)).to(gpu_device),
                        "label_det": torch.from_numpy(synthetic.det_labels(B, gen=gen, height=S, width=S)).to(gpu_device),
                        "label_seg": torch.from_numpy(synthetic.seg_labels(B, S, S, gen=gen)).to(gpu_device),
                        "disparity": (gen.random((B, 64, 128)) * 6000 + 800).astype(np.float32)})
    cls = ["c%d" % i for i in range(8)]
    seg = ["s%d" % i for i in range(19)]
    out = evaluate_net(net, batches, cls, seg, full_res=(256, 256))
    for k in ("CrossEntropy", "SmoothL1", "accuracy", "mAP", "mIoU"):
        assert np.isfinite(out[k]), (k, out[k])
    assert "derror" in out and len(out["class_maps"]) == 2 and out["class_maps"][0].shape == (B, 256, 256)
    # the last batch is still in the graph: its read-outs against the restatements on the same device outputs
    det = net.det.out.data
    pred = filter_detections(det)
    host = det.cpu().numpy()
    for b in range(B):
        rows = host[b][host[b][:, 0] >= 0]
        rows = rows[rows[:, 1] > 0.1]
        np.testing.assert_array_equal(pred[b, :rows.shape[0]], rows)
        assert (pred[b, rows.shape[0]:, 0] == -1).all()
    prob = fn.nhwc_to_nchw(net.seg_out.prob.data, 19).cpu().numpy()
    np.testing.assert_array_equal(out["class_maps"][1].cpu().numpy(), om.upsample_argmax(prob, 256, 256))
    ids = label_ids(out["class_maps"][1]).cpu().numpy()
    assert set(np.unique(ids)).issubset({7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33})
