"""Segmentation readouts (the two that are pixel-parallel): device counts vs the numpy restatement."""
import numpy as np
import pytest

from oracle import metrics as om


def test_oracle_hand_case():
    label = np.array([[[0, 1], [1, 255]]], np.float32)                      # (B=1, 2, 2)
    pred = np.zeros((1, 3, 2, 2), np.float32)
    pred[0, 0, 0, 0] = 1; pred[0, 1, 0, 1] = 1; pred[0, 2, 1, 0] = 1; pred[0, 1, 1, 1] = 1   # argmax: 0 1 / 2 1
    assert om.custom_accuracy_update(label, pred) == (2, 4)                 # the 255 pixel counts as a miss
    inter, total = om.iou_update(label, pred, 4)
    assert inter.tolist() == [1, 1, 0, 0] and total.tolist() == [1, 3, 1, 0]
    vals = om.iou_get(inter, total)
    assert abs(vals[-1] - np.mean([1 / (1 + 1e-5), 1 / (3 + 1e-5), 0.0])) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["NCHW", "NHWC"])
def test_device_counts_match_restatement(gpu_device, layout):
    import torch
    from dspnet_amd.train.metric import CustomAccuracyMetric, IoUMetric
    g = np.random.Generator(np.random.PCG64(5))
    B, C, H, W = 3, 19, 37, 41
    pred = g.standard_normal((B, C, H, W)).astype(np.float32)
    pred[0, :, 0, 0] = 0.5                                                   # all-equal scores: first index wins
    label = g.integers(0, C, (B, H, W)).astype(np.float32)
    label[g.random(label.shape) < 0.1] = 255
    names = ["c%d" % i for i in range(C)]
    acc, iou = CustomAccuracyMetric(num_classes=C), IoUMetric(class_names=names)
    if layout == "NCHW":
        pd = torch.from_numpy(pred).cuda()
    else:
        pd = torch.zeros(B, H, W, 20); pd[..., :C] = torch.from_numpy(pred).permute(0, 2, 3, 1); pd = pd.cuda()
    ld = torch.from_numpy(label).cuda()
    for _ in range(2):                                                       # accumulation over two updates
        acc.update([ld], [pd]); iou.update([ld], [pd])
    s, n = om.custom_accuracy_update(label, pred)
    assert acc.get() == ("accuracy", (2 * s) / (2 * n))
    inter, total = om.iou_update(label, pred, C + 1)
    ref = om.iou_get(2 * inter, 2 * total)
    nm, vals = iou.get()
    assert nm == names + ["mIoU"]
    np.testing.assert_allclose(vals, ref, rtol=0, atol=1e-12)
