"""Detection read-outs of the reference's evaluation script (evaluate/eval_metric.py).

MApMetric / VOC07MApMetric mirror evaluate/eval_metric.py:4-276: same constructor, `update(labels, preds)`,
`get()` and `reset()` contracts, same record keeping (score, 1 = true positive / 2 = false positive per class,
ground-truth counts per class) and the same average-precision integrals.  They are host code in the reference
(numpy loops over the few hundred boxes that survive NMS) and stay host code here; the device work of the evaluation
(forward pass, detection operator, segmentation counts, full-resolution argmax) is in the HIP library.

Behaviour kept on purpose (it is what the reference computes, see the cited lines):
  * detections are NOT re-sorted by score inside `update` (:123 sorts into a temporary that is dropped); the
    matching runs in the order the detection operator emitted them;
  * class ids are truncated toward zero (`astype(int)`), so an id in (-1, 0) counts as class 0 (:113-118);
  * a 6th label column is a "difficult" flag: with use_difficult=False a match to a ground truth whose flag is > 0
    is dropped, and only flags < 1 count as ground truths (:137-140, :155-158).  multi_eval.py:376 slices the
    labels to 5 columns before the call, so the depth column is never read as that flag;
  * a class whose detections all matched difficult boxes adds nothing, not even its ground-truth count (:163-165);
  * ground truths of classes without detections add the record [0, 0] and their plain count (:168-175).

IoUMetric (evaluate/eval_metric.py:278-388) lives in dspnet_amd.train.metric (device counts) and is re-exported."""
import numpy as np

from ..train.metric import IoUMetric  # noqa: F401  (same module path as the reference: evaluate.eval_metric.IoUMetric)


def _np(a):
    """torch tensor (any device) / numpy / list -> numpy float32 (the reference's NDArray.asnumpy())"""
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float32)


def _box_iou(box, others):
    """IoU of one box with rows of `others` (evaluate/eval_metric.py:83-107); unions below 1e-12 give 0"""
    iw = np.maximum(np.minimum(others[:, 2], box[2]) - np.maximum(others[:, 0], box[0]), 0.)
    ih = np.maximum(np.minimum(others[:, 3], box[3]) - np.maximum(others[:, 1], box[1]), 0.)
    inter = iw * ih
    union = (box[2] - box[0]) * (box[3] - box[1]) + (others[:, 2] - others[:, 0]) * (others[:, 3] - others[:, 1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = inter / union
    iou[union < 1e-12] = 0
    return iou


def _first_seen(ids):
    """distinct values of an int array in order of first appearance"""
    _, first = np.unique(ids, return_index=True)
    return ids[np.sort(first)]


class MApMetric(object):
    """mean average precision over classes (evaluate/eval_metric.py:4-253)

    ovp_thresh: IoU above which a detection matches a ground truth; use_difficult: count "difficult" ground truths;
    class_names: optional list of str -> get() returns one AP per class plus 'mAP'; pred_idx: index into `preds`."""

    def __init__(self, ovp_thresh=0.5, use_difficult=False, class_names=None, pred_idx=0):
        self.name = "mAP"
        if class_names is None:
            self.num = None
        else:
            assert isinstance(class_names, (list, tuple))
            for name in class_names:
                assert isinstance(name, str), "must provide names as str"
            self.name = list(class_names) + ["mAP"]
            self.num = len(class_names) + 1
        self.reset()
        self.ovp_thresh = ovp_thresh
        self.use_difficult = use_difficult
        self.class_names = class_names
        self.pred_idx = int(pred_idx)

    def reset(self):
        if getattr(self, "num", None) is None:
            self.num_inst, self.sum_metric = 0, 0.0
        else:
            self.num_inst, self.sum_metric = [0] * self.num, [0.0] * self.num
        self.records = dict()   # class id -> (k, 2) array [score, 1 tp | 2 fp]
        self.counts = dict()    # class id -> number of ground truths

    # ---- update: records only; the integrals are taken in get() -------------------------------------------------
    def update(self, labels, preds):
        """labels[0]: (B, n, 5|6) rows [id, xmin, ymin, xmax, ymax, (difficult)], id < 0 = padding;
        preds[pred_idx]: (B, m, 6) rows [id, score, xmin, ymin, xmax, ymax], id < 0 = empty slot"""
        all_labels, all_preds = _np(labels[0]), _np(preds[self.pred_idx])
        for label, pred in zip(all_labels, all_preds):
            self._update_image(label, pred)

    def _update_image(self, label, pred):
        pred_ids = pred[:, 0].astype(int)
        label_ids = label[:, 0].astype(int)
        label_open = np.ones(label.shape[0], bool)          # ground truths no detection class has claimed yet
        has_flag = label.shape[1] >= 6
        for cid in _first_seen(pred_ids):
            if cid < 0:
                continue
            dets = pred[pred_ids == cid]                    # emission order (see the module note)
            gt_rows = np.where(label_open & (label_ids == cid))[0]
            gts = label[gt_rows]
            label_open[gt_rows] = False
            flags = np.full(dets.shape[0], 2.0)             # no ground truth of this class: all false positives
            if gts.shape[0] > 0:
                taken = np.zeros(gts.shape[0], bool)
                for j in range(dets.shape[0]):
                    ious = _box_iou(dets[j, 2:], gts[:, 1:5])
                    k = int(np.argmax(ious))
                    if not ious[k] > self.ovp_thresh:
                        flags[j] = 2
                    elif not self.use_difficult and has_flag and gts[k, 5] > 0:
                        flags[j] = 0                        # matched a difficult box: neither tp nor fp
                    elif taken[k]:
                        flags[j] = 2                        # duplicate
                    else:
                        flags[j] = 1
                        taken[k] = True
            if not self.use_difficult and has_flag:
                gt_count = int(np.sum(gts[:, 5] < 1))
            else:
                gt_count = gts.shape[0]
            keep = flags > 0
            if keep.any():
                self._insert(int(cid), np.stack([dets[keep, 1].astype(np.float64), flags[keep]], axis=1), gt_count)
        # ground truths of classes that were not predicted at all
        rest_ids = label_ids[label_open]
        for cid in _first_seen(rest_ids):
            if cid < 0:
                continue
            self._insert(int(cid), np.array([[0., 0.]]), int(np.sum(rest_ids == cid)))

    def _insert(self, key, records, count):
        if key not in self.records:
            assert key not in self.counts
            self.records[key] = records
            self.counts[key] = count
        else:
            self.records[key] = np.vstack((self.records[key], records))
            self.counts[key] += count

    # ---- get ----------------------------------------------------------------------------------------------------
    def get(self):
        self._update()
        if self.num is None:
            if self.num_inst == 0:
                return (self.name, float("nan"))
            return (self.name, self.sum_metric / self.num_inst)
        names = ["%s" % (self.name[i]) for i in range(self.num)]
        values = [x / y if y != 0 else float("nan") for x, y in zip(self.sum_metric, self.num_inst)]
        return (names, values)

    def _update(self):
        aps = []
        for k, v in self.records.items():
            recall, prec = self._recall_prec(v, self.counts[k])
            ap = self._average_precision(recall, prec)
            aps.append(ap)
            if self.num is not None and k < (self.num - 1):
                self.sum_metric[k] = ap
                self.num_inst[k] = 1
        mean_ap = np.mean(aps) if aps else float("nan")     # (np.mean([]) in the reference: nan with a warning)
        if self.num is None:
            self.num_inst, self.sum_metric = 1, mean_ap
        else:
            self.num_inst[-1], self.sum_metric[-1] = 1, mean_ap

    def _recall_prec(self, record, count):
        """cumulative recall / precision over the records sorted by descending score (:196-207)"""
        record = record[record[:, 1].astype(int) != 0]
        ranked = record[record[:, 0].argsort()[::-1]]
        tp = np.cumsum(ranked[:, 1].astype(int) == 1)
        fp = np.cumsum(ranked[:, 1].astype(int) == 2)
        recall = tp * 0.0 if count <= 0 else tp / float(count)
        prec = tp.astype(float) / (tp + fp)
        return recall, prec

    def _average_precision(self, rec, prec):
        """area under the monotone precision envelope (:209-238)"""
        mrec = np.concatenate(([0.], rec, [1.]))
        mpre = np.concatenate(([0.], prec, [0.]))
        mpre = np.maximum.accumulate(mpre[::-1])[::-1]      # envelope: running maximum from the right
        step = np.where(mrec[1:] != mrec[:-1])[0]
        return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


class VOC07MApMetric(MApMetric):
    """PASCAL VOC 2007 11-point interpolated AP (evaluate/eval_metric.py:255-276)"""

    def _average_precision(self, rec, prec):
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            above = rec >= t
            p = np.max(prec[above]) if np.sum(above) != 0 else 0
            ap += p / 11.
        return ap
