"""numpy restatement of the reference's evaluation readouts (segmentation counts, detection mAP, distance error,
full-resolution class map).  TEST INFRASTRUCTURE ONLY.
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


# ---------------------------------------------------------------------------------------------------------------
# Detection mAP (evaluate/eval_metric.py:4-276), restated in the reference's own control flow (np.delete loops)
# ---------------------------------------------------------------------------------------------------------------
class MApOracle(object):
    def __init__(self, ovp_thresh=0.5, use_difficult=False, num_classes=None, voc07=False):
        self.ovp_thresh, self.use_difficult, self.voc07 = ovp_thresh, use_difficult, voc07
        self.num = None if num_classes is None else num_classes + 1
        self.records, self.counts = dict(), dict()

    @staticmethod
    def iou(x, ys):
        """:83-107"""
        ixmin = np.maximum(ys[:, 0], x[0]); iymin = np.maximum(ys[:, 1], x[1])
        ixmax = np.minimum(ys[:, 2], x[2]); iymax = np.minimum(ys[:, 3], x[3])
        iw = np.maximum(ixmax - ixmin, 0.); ih = np.maximum(iymax - iymin, 0.)
        inters = iw * ih
        uni = (x[2] - x[0]) * (x[3] - x[1]) + (ys[:, 2] - ys[:, 0]) * (ys[:, 3] - ys[:, 1]) - inters
        with np.errstate(divide="ignore", invalid="ignore"):
            ious = inters / uni
        ious[uni < 1e-12] = 0
        return ious

    def update(self, labels, preds):
        """:69-175; labels (B, n, 5|6) float32, preds (B, m, 6) float32"""
        for i in range(labels.shape[0]):
            label = labels[i].astype(np.float32); pred = preds[i].astype(np.float32)
            while pred.shape[0] > 0:
                cid = int(pred[0, 0])
                indices = np.where(pred[:, 0].astype(int) == cid)[0]
                if cid < 0:
                    pred = np.delete(pred, indices, axis=0)
                    continue
                dets = pred[indices]
                pred = np.delete(pred, indices, axis=0)
                dets[dets[:, 1].argsort()[::-1]]   # (:123) result dropped: no re-ordering
                records = np.hstack((dets[:, 1][:, np.newaxis], np.zeros((dets.shape[0], 1))))
                label_indices = np.where(label[:, 0].astype(int) == cid)[0]
                gts = label[label_indices, :]
                label = np.delete(label, label_indices, axis=0)
                if gts.size > 0:
                    found = [False] * gts.shape[0]
                    for j in range(dets.shape[0]):
                        ious = self.iou(dets[j, 2:], gts[:, 1:5])
                        ovargmax = np.argmax(ious)
                        ovmax = ious[ovargmax]
                        if ovmax > self.ovp_thresh:
                            if not self.use_difficult and gts.shape[1] >= 6 and gts[ovargmax, 5] > 0:
                                pass
                            elif not found[ovargmax]:
                                records[j, -1] = 1
                                found[ovargmax] = True
                            else:
                                records[j, -1] = 2
                        else:
                            records[j, -1] = 2
                else:
                    records[:, -1] = 2
                if not self.use_difficult and gts.shape[1] >= 6:
                    gt_count = np.sum(gts[:, 5] < 1)
                else:
                    gt_count = gts.shape[0]
                records = records[np.where(records[:, -1] > 0)[0], :]
                if records.size > 0:
                    self._insert(cid, records, gt_count)
            while label.shape[0] > 0:
                cid = int(label[0, 0])
                label_indices = np.where(label[:, 0].astype(int) == cid)[0]
                label = np.delete(label, label_indices, axis=0)
                if cid < 0:
                    continue
                self._insert(cid, np.array([[0, 0]]), label_indices.size)

    def _insert(self, key, records, count):
        if key not in self.records:
            self.records[key] = records; self.counts[key] = count
        else:
            self.records[key] = np.vstack((self.records[key], records)); self.counts[key] += count

    def get(self):
        """:177-193 + :46-67 -> list of values (per class + mean when num is set, else [mean])"""
        aps = []
        per = {}
        for k, v in self.records.items():
            record = np.delete(v, np.where(v[:, 1].astype(int) == 0)[0], axis=0)
            srt = record[record[:, 0].argsort()[::-1]]
            tp = np.cumsum(srt[:, 1].astype(int) == 1); fp = np.cumsum(srt[:, 1].astype(int) == 2)
            recall = tp * 0.0 if self.counts[k] <= 0 else tp / float(self.counts[k])
            prec = tp.astype(float) / (tp + fp)
            ap = self._ap07(recall, prec) if self.voc07 else self._ap(recall, prec)
            aps.append(ap); per[k] = ap
        if self.num is None:
            return [np.mean(aps)]
        vals = [per.get(k, float("nan")) if k in per else float("nan") for k in range(self.num - 1)]
        return vals + [np.mean(aps)]

    @staticmethod
    def _ap(rec, prec):
        """:209-238"""
        mrec = np.concatenate(([0.], rec, [1.])); mpre = np.concatenate(([0.], prec, [0.]))
        for i in range(mpre.size - 1, 0, -1):
            mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
        i = np.where(mrec[1:] != mrec[:-1])[0]
        return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])

    @staticmethod
    def _ap07(rec, prec):
        """:260-276"""
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap += p / 11.
        return ap


def distance_errors(disparities, dets, num_classes):
    """train/metric.py:190-231 for one update call -> list (per class) of relative errors.  disparities (B, hh, ww),
    dets: list of (n, N, 7).  Python-2 arithmetic of the reference written out: roi.shape[0]/2 is a floor division,
    numpy-1.x scalar promotion (float32 scalar with a Python float -> float64).  One-pixel boxes (a crash in the
    reference: np.sort of a 0-d array) are evaluated."""
    import math
    _, hh, ww = disparities.shape
    error = [[] for _ in range(num_classes)]
    for disparity, imgs in zip(disparities, dets):
        for img in imgs:
            for bbox in img.astype(np.float32):
                if bbox[0] < 0:
                    break
                xmin, xmax = int(bbox[2] * ww), int(bbox[4] * ww)
                ymin, ymax = int(bbox[3] * hh), int(bbox[5] * hh)
                xmin, ymin = max(0, xmin), max(0, ymin)
                if xmin == xmax:
                    xmax = xmin + 1
                roi = disparity[ymin:ymax, xmin:xmax]
                roi = np.atleast_1d(np.squeeze(roi.reshape((1, -1)))).astype(np.float32)
                roi = np.sort(roi)
                if roi.shape[0] == 0:
                    continue
                dist = 2200. * 75. / (np.float64(roi[int(math.ceil(roi.shape[0] // 2))]) + 1e-3)
                if dist > 1000:
                    dist = 200
                if dist > 199:
                    continue
                error[int(bbox[0])].append(math.fabs(np.float64(bbox[6]) * 255. - dist) / dist)
    return error


def upsample_argmax(prob, Ho, Wo):
    """multi_eval.py:28-34 for prob (N, C, h, w) float32 -> uint8 (N, Ho, Wo): GridGenerator(affine identity) ->
    BilinearSampler -> argmax(axis=1).  fp32 arithmetic in the sampler's order
    tl*wy*wx + tr*wy*(1-wx) + bl*(1-wy)*wx + br*(1-wy)*(1-wx); corners outside the map count as 0."""
    N, C, h, w = prob.shape
    f = np.float32

    def coords(O, I):
        o = np.arange(O, dtype=np.float32)
        g = (f(-1.) + o * (f(2.) / f(O - 1))) if O > 1 else np.zeros(O, np.float32)
        s = (g + f(1.)) * f(I - 1) / f(2.)
        i0 = np.floor(s).astype(np.int64)
        w0 = f(1.) - (s - i0.astype(np.float32))
        return i0, w0.astype(np.float32)

    y0, wy = coords(Ho, h)
    x0, wx = coords(Wo, w)

    def gather(yy, xx):
        ok = ((yy >= 0) & (yy < h))[:, None] & ((xx >= 0) & (xx < w))[None, :]
        v = prob[:, :, np.clip(yy, 0, h - 1)][:, :, :, np.clip(xx, 0, w - 1)]
        return np.where(ok[None, None], v, f(0.))

    WY, WX = wy[:, None], wx[None, :]
    one = f(1.)
    out = gather(y0, x0) * WY * WX + gather(y0, x0 + 1) * WY * (one - WX) \
        + gather(y0 + 1, x0) * (one - WY) * WX + gather(y0 + 1, x0 + 1) * (one - WY) * (one - WX)
    assert out.dtype == np.float32
    return np.argmax(out, axis=1).astype(np.uint8)
