"""Loss readouts of the reference, computed on the device buffers.

MultiBoxMetric.update (train/metric.py:27-46):
    CrossEntropy = sum_{label>=0} -log(cls_prob[label] + 1e-8) / #(cls_label >= 0)
    SmoothL1     = sum(loc_loss) / #(cls_label >= 0)
plus the segmentation cross-entropy over labels != 255 (the quantity SoftmaxOutput(seg) descends)."""
from .. import functional as fn


class MultiBoxMetric:
    def __init__(self, eps=1e-8):
        self.eps = eps
        self.reset()

    def reset(self):
        self.num = 2
        self.sum_metric = [0.0, 0.0, 0.0]
        self.num_inst = [0, 0, 0]

    def update(self, net):
        """reads the device buffers of a MultiTaskNet after forward(); one small D2H copy"""
        if net.cls_out is not None:
            B, C, N = net.cls_out.cls_prob.shape
            ce = fn.cross_entropy_sum(net.cls_out.prob_nc.view(B * N, C), net.target.cls_target, C, -1.0, self.eps)
            sl1 = fn.sum_all(net.loc_loss.out.data)
        else:   # segmentation-only graph
            import torch
            ce, sl1 = torch.zeros(2), torch.zeros(1)
        if net.seg_out is not None:
            sp = net.seg_out.prob.data
            rows = sp.numel() // sp.shape[-1]
            seg = fn.cross_entropy_sum(sp.view(rows, sp.shape[-1]), net.label_seg.data, net.seg_out.C, 255.0,
                                       self.eps).cpu()
        else:   # detection-only graph
            seg = [0.0, 0.0]
        ce, sl1 = ce.cpu(), sl1.cpu()
        valid = float(ce[1])
        self.sum_metric[0] += float(ce[0]); self.num_inst[0] += valid
        self.sum_metric[1] += float(sl1[0]); self.num_inst[1] += valid
        self.sum_metric[2] += float(seg[0]); self.num_inst[2] += float(seg[1])

    def get(self):
        names = ["CrossEntropy", "SmoothL1", "SegCrossEntropy"]
        vals = [s / n if n > 0 else float("nan") for s, n in zip(self.sum_metric, self.num_inst)]
        return names, vals


def _seg_inputs(labels, preds, num_classes):
    """(scores (rows, ld) NHWC-style device tensor, labels) from either this build's NHWC probabilities
    (B, H, W, ld) or the reference's layout (B, C, H, W)"""
    import torch
    from .. import functional as fn
    if preds.dim() == 4 and tuple(preds.shape[2:]) == tuple(labels.shape[1:]) and preds.shape[1] >= num_classes:
        preds = fn.nchw_to_nhwc(preds.contiguous(), Cp=fn.pad4(preds.shape[1]))      # (B, C, H, W) -> (B, H, W, Cp)
    return preds.contiguous(), labels.contiguous().to(torch.float32)


class CustomAccuracyMetric:
    """Pixel accuracy of the segmentation output (train/metric.py:71-133): pred = argmax over the class axis,
    sum_metric += #(pred == label), num_inst += #pixels (ignore-labelled pixels count as misses, as in the reference).
    The counting runs on the device (dspn_seg_counts_f32)."""

    def __init__(self, axis=1, name="accuracy", num_classes=19):
        self.axis, self.name, self.num_classes = axis, name, num_classes
        self.reset()

    def reset(self):
        self.sum_metric, self.num_inst = 0.0, 0

    def update(self, labels, preds):
        from .. import functional as fn
        for label, pred in zip(labels, preds):
            scores, lab = _seg_inputs(label, pred, self.num_classes)
            c = fn.seg_counts(scores, lab, self.num_classes).cpu()
            self.sum_metric += float(c[3 * self.num_classes])
            self.num_inst += lab.numel()

    def get(self):
        return self.name, (self.sum_metric / self.num_inst if self.num_inst else float("nan"))


class IoUMetric:
    """Per-class intersection / union counts and their mean (evaluate/eval_metric.py:278-388):
    sum_metric[c] += #(label == c & pred == c), num_inst[c] += #(label == c | pred == c);
    get(): mIoU = mean(sum_metric[:-1] / (num_inst[:-1] + 1e-5)) (:386-388)."""

    def __init__(self, axis=1, name="mIoU", class_names=None):
        assert isinstance(class_names, (list, tuple)) and all(isinstance(n, str) for n in class_names)
        self.axis = axis
        self.class_names = list(class_names)
        self.name = self.class_names + [name]
        self.num = len(self.class_names) + 1
        self.reset()

    def reset(self):
        import numpy as np
        self.num_inst = np.zeros(self.num)
        self.sum_metric = np.zeros(self.num)

    def update(self, labels, preds):
        from .. import functional as fn
        C = self.num - 1
        for label, pred in zip(labels, preds):
            scores, lab = _seg_inputs(label, pred, C)
            c = fn.seg_counts(scores, lab, C).cpu().numpy().astype("float64")
            inter, npred, nlab = c[:C], c[C:2 * C], c[2 * C:3 * C]
            self.sum_metric[:C] += inter
            self.num_inst[:C] += npred + nlab - inter
            # the reference also loops idx == C (the mIoU slot): no pixel carries that id, both sums stay 0

    def get(self):
        import numpy as np
        self.sum_metric[-1] = np.mean(self.sum_metric[:-1] / (self.num_inst[:-1] + 1e-5))
        self.num_inst[-1] = 1.0
        names = ["%s" % n for n in self.name]
        values = [x / y if y != 0 else float("nan") for x, y in zip(self.sum_metric, self.num_inst)]
        return names, values


class DistanceAccuracyMetric:
    """Relative error of the predicted object distance against the disparity map (train/metric.py:135-260).

    update(labels, preds): labels (B, hh, ww) disparity maps (host array or tensor), preds = list of detection
    tensors (n, N, 7) rows [id, score, xmin, ymin, xmax, ymax, dist], paired sample-wise with the disparity maps
    exactly as the reference's `zip(labels, preds)` pairs them.  For every detection up to the first id < 0:
    pixel box (truncating int(), xmin / ymin clamped at 0, an empty column range widened to 1 pixel), reference
    distance = 2200 * 75 / (q + 1e-3) with q the element of rank n // 2 of the sorted disparities in the box
    (`int(math.ceil(n / 2))` under Python 2 integer division), > 1000 -> 200, > 199 -> skipped; the error
    |dist_pred * 255 - dist| / dist is collected per class.  get(): per-class mean and the overall mean ('derror').

    Host code, as in the reference (a few boxes per image).  Differences from the reference, both deliberate:
    a one-pixel box is evaluated (np.squeeze makes it 0-d there and np.sort raises), and get() only writes
    "dist_errors.txt" when dump_errors is given."""

    def __init__(self, class_names, name="derror", dump_errors=None):
        self.name = list(class_names) + [name]
        self.num = len(class_names) + 1
        self.dump_errors = dump_errors
        self.reset()

    def reset(self):
        self.num_inst = [0] * self.num
        self.sum_metric = [0.0] * self.num
        self.errors = []

    @staticmethod
    def _host(a):
        import numpy as np
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        return np.asarray(a)

    def update(self, labels, preds):
        import math
        import numpy as np
        labels = self._host(labels)
        _, hh, ww = labels.shape
        error = [[] for _ in range(self.num - 1)]
        for disparity, dets in zip(labels, preds):
            dets = self._host(dets).astype(np.float32, copy=False)
            for img in dets.reshape((-1,) + dets.shape[-2:]):
                for bbox in img:
                    if bbox[0] < 0:
                        break
                    xmin, xmax = int(bbox[2] * np.float32(ww)), int(bbox[4] * np.float32(ww))
                    ymin, ymax = int(bbox[3] * np.float32(hh)), int(bbox[5] * np.float32(hh))
                    xmin, ymin = max(0, xmin), max(0, ymin)
                    if xmin == xmax:
                        xmax = xmin + 1
                    roi = disparity[ymin:ymax, xmin:xmax].reshape(-1).astype(np.float32)
                    if roi.shape[0] == 0:
                        continue
                    k = roi.shape[0] // 2
                    q = float(np.partition(roi, k)[k])      # rank-k element == np.sort(roi)[k]
                    dist = 2200. * 75. / (q + 1e-3)
                    if dist > 1000:
                        dist = 200
                    if dist > 199:
                        continue
                    error[int(bbox[0])].append(math.fabs(float(bbox[6]) * 255. - dist) / dist)
        for i in range(self.num - 1):
            self.sum_metric[i] += math.fsum(error[i])
            self.num_inst[i] += len(error[i])
            self.errors += error[i]
        self.sum_metric[self.num - 1] += math.fsum([math.fsum(e) for e in error])
        self.num_inst[self.num - 1] += math.fsum([len(e) for e in error])

    def get(self):
        names = ["%s" % n for n in self.name]
        values = [x / y if y != 0 else float("nan") for x, y in zip(self.sum_metric, self.num_inst)]
        if self.dump_errors:
            import numpy as np
            np.savetxt(self.dump_errors, np.array(self.errors) * 100., fmt="%.1f")
        return names, values
