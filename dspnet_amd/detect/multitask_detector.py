"""Inference wrapper: what `Detector` feeds and reads in the reference.

detect/multitask_detector.py:99-163 loads the saved TRAINING symbol, binds zero `label_det (B,200,6)` and
`seg_out_label`, runs forward(is_train=True) per image and reads outputs[3] (det_out) and outputs[4]
(seg probabilities); rows with id >= 0 are detections (:268-271) and the seg map is the arg-max over
the class axis (:263).  det_out does not depend on MultiBoxTarget, so this forward-only path runs the
test graph (`get_multi_symbol`): same det / seg values, no target / loss kernels.  Image decoding,
resizing and drawing (cv2) are outside the hot path."""
import torch

from .. import functional as fn
from ..symbol.multitask_symbol_factory import get_multi_symbol

MEAN_RGB = (123.0, 117.0, 104.0)


class Detector:
    def __init__(self, network="resnet-50", data_shape=512, num_classes=8, batch_size=1, mean_pixels=MEAN_RGB,
                 nms_thresh=0.5, force_suppress=False, nms_topk=400, device=None, params=None, seed=0,
                 model_prefix=None, epoch=0):
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.net = get_multi_symbol(network, data_shape, num_classes=num_classes, batch_size=batch_size,
                                    nms_thresh=nms_thresh, force_suppress=force_suppress, nms_topk=nms_topk,
                                    device=self.device, seed=seed)
        if params:
            self.net.g.load_params(params)
        if model_prefix is not None:   # mx.model.load_checkpoint(model_prefix, epoch) (detect/multitask_detector.py:105)
            from ..model import load_checkpoint
            _, args, _ = load_checkpoint(model_prefix, epoch)
            self.net.g.set_params(args)
        self.mean = torch.tensor(mean_pixels, dtype=torch.float32, device=self.device).view(1, 3, 1, 1)

    def forward(self, data=None):
        """data: (B,3,H,W) float32 device tensor, RGB, mean already subtracted (dataset/iterator.py:570-571)"""
        if data is not None:
            self.net.data.data.copy_(data)
        g = self.net.g
        if g.scalars is not None and g.guard["enabled"] and not g.guard["have_stats"]:
            # range guard of the default math (advisor r5): the very first batch of a freshly loaded net has no spans to decide
            # from -- one extra forward measures them (inputs, weights), the next decides from that pass before it runs
            g.forward()
            g.guard["decide_now"] = True
        g.forward()
        self.net.det.join()        # MultiBoxDetection runs on a side stream beside the seg decoder
        return self.net.det.out.data, self.net.seg_out.prob.data

    def detect(self, data, thresh=0.0):
        """-> (list of per-image (k,7) tensors [id, score, xmin, ymin, xmax, ymax, dist] with id >= 0 and
        score > thresh, seg probabilities (B,19,H/4,W/4))"""
        det, _ = self.forward(data)
        det = det.cpu()
        out = []
        for b in range(det.shape[0]):
            rows = det[b]
            out.append(rows[(rows[:, 0] >= 0) & (rows[:, 1] > thresh)])
        return out, fn.nhwc_to_nchw(self.net.seg_out.prob.data, self.net.seg_out.C)
