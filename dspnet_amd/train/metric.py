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
        B, C, N = net.cls_out.cls_prob.shape
        ce = fn.cross_entropy_sum(net.cls_out.prob_nc.view(B * N, C), net.target.cls_target, C, -1.0, self.eps)
        sl1 = fn.sum_all(net.loc_loss.out.data)
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
