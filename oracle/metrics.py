"""numpy restatement of the reference's segmentation readouts.  TEST INFRASTRUCTURE ONLY.
PARITY STATUS: "parity unpinned" (no vectors in the reference; MXNet's metric base class is not available, the
update arithmetic of train/metric.py:100-133 and evaluate/eval_metric.py:359-388 is restated line by line)."""
import numpy as np


def custom_accuracy_update(label, pred):
    """train/metric.py:116-133 -> (sum_metric increment, num_inst increment); pred (B, C, H, W) scores"""
    if pred.shape != label.shape:
        pred = np.argmax(pred, axis=1)
    pred = pred.astype("int32"); label = label.astype("int32")
    return int((pred.flat == label.flat).sum()), int(pred.size)


def iou_update(label, pred, num):
    """evaluate/eval_metric.py:359-384 with self.num = num (= classes + 1) -> (inter[num], total[num])"""
    if pred.shape != label.shape:
        pred = np.argmax(pred, axis=1)
    pred = pred.astype("int32"); label = label.astype("int32")
    inter = np.zeros(num); total = np.zeros(num)
    for idx in range(num):
        inter[idx] = ((label.flat == idx) & (pred.flat == idx)).sum()
        total[idx] = ((label.flat == idx) | (pred.flat == idx)).sum()
    return inter, total


def iou_get(sum_metric, num_inst):
    """evaluate/eval_metric.py:337-357, 386-388"""
    sum_metric = sum_metric.copy(); num_inst = num_inst.copy()
    sum_metric[-1] = np.mean(sum_metric[:-1] / (num_inst[:-1] + 1e-5))
    num_inst[-1] = 1.0
    return [x / y if y != 0 else float("nan") for x, y in zip(sum_metric, num_inst)]
