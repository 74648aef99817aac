"""Evaluation loop of multi_eval.py (evaluate_net, :156-425) over this build's graph.

What the reference does per batch, and where it runs here:
  forward of the TRAINING symbol with is_train=True (:312)                 -> net.g.forward() (HIP graph)
  MultiBoxMetric over cls_prob / loc_loss / cls_label (:372-374)           -> train.metric.MultiBoxMetric (device sums)
  CustomAccuracyMetric + IoUMetric over seg_out (:375, :378)               -> dspn_seg_counts_f32 (device counts)
  detections with id >= 0 and score > .1 (:329-335), MApMetric (:376-377)  -> host, a few hundred rows
  seg probabilities upsampled to 1024x2048 + argmax (:28-34, :355)         -> dspn_seg_upsample_argmax_f32 (fused)
  DistanceAccuracyMetric against the disparity maps (:379-384)             -> host, as in the reference
The image display / file writing of the script (cv2) is not part of the numerics contract and is not built."""
import numpy as np
import torch

from .. import functional as fn
from ..train.metric import CustomAccuracyMetric, DistanceAccuracyMetric, IoUMetric, MultiBoxMetric
from .eval_metric import MApMetric, VOC07MApMetric

# trainId -> labelId table the script applies before writing result images (multi_eval.py:352-353)
CITYSCAPES_LABEL_IDS = (7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33)


def prob_upsampling(seg_prob, target_shape=(1024, 2048), num_classes=19):
    """multi_eval.py:28-34: class map (uint8) of the probabilities sampled bilinearly at target_shape.
    seg_prob: device tensor, NHWC (B, h, w, ld) as the graph holds it, or NCHW (C, h, w) / (B, C, h, w) as the
    reference passes it.  Returns (B, H, W) uint8 on the device (squeezed to (H, W) for a single 3-d input)."""
    squeeze = seg_prob.dim() == 3
    if squeeze:
        seg_prob = seg_prob.unsqueeze(0)
    if seg_prob.shape[1] == num_classes and seg_prob.shape[-1] != fn.pad4(num_classes):        # NCHW
        seg_prob = fn.nchw_to_nhwc(seg_prob.contiguous(), Cp=fn.pad4(num_classes))
    out = fn.seg_upsample_argmax(seg_prob.contiguous(), num_classes, int(target_shape[0]), int(target_shape[1]))
    return out[0] if squeeze else out


def label_ids(class_map):
    """trainId map -> Cityscapes labelId map (the cv2.LUT of multi_eval.py:352-356), on the device"""
    lut = torch.zeros(256, dtype=torch.uint8, device=class_map.device)
    lut[:19] = torch.tensor(CITYSCAPES_LABEL_IDS, dtype=torch.uint8, device=class_map.device)
    return lut[class_map.long()]


def filter_detections(det, score_thresh=0.1):
    """multi_eval.py:329-335: keep rows with id >= 0, then score > score_thresh.  det (B, N, 7) device or host.
    -> host float32 (B, kmax, 7), short images padded with -1 rows (B = 1 reproduces the reference's array)."""
    det = det.detach().cpu().numpy() if hasattr(det, "detach") else np.asarray(det)
    rows = [d[(d[:, 0] >= 0)] for d in det]
    rows = [d[d[:, 1] > score_thresh] for d in rows]
    kmax = max([r.shape[0] for r in rows] + [0])
    out = np.full((det.shape[0], kmax, det.shape[2]), -1.0, np.float32)
    for b, r in enumerate(rows):
        out[b, :r.shape[0]] = r
    return out


def evaluate_net(net, batches, class_names, seg_class_names, ovp_thresh=0.5, use_difficult=False,
                 voc07_metric=False, full_res=None, score_thresh=0.1):
    """net: training graph (symbol.multitask_symbol_factory.get_multi_symbol_train); batches: iterable of dicts with
    'data' (B,3,H,W), 'label_det' (B,L,6), 'label_seg' (B,H/4,W/4) and optionally 'disparity' (B,hh,ww) host maps.
    -> dict name -> value, plus 'class_maps' (list of uint8 device tensors) when full_res=(H, W) is given."""
    multibox_metric = MultiBoxMetric()
    acc_metric = CustomAccuracyMetric(num_classes=len(seg_class_names))
    depth_metric = DistanceAccuracyMetric(class_names=list(class_names))
    det_metric = (VOC07MApMetric if voc07_metric else MApMetric)(ovp_thresh, use_difficult, list(class_names))
    seg_metric = IoUMetric(class_names=list(seg_class_names), axis=1)
    class_maps = []
    for batch in batches:
        net.data.data.copy_(batch["data"])
        net.label_det.data.copy_(batch["label_det"])
        net.label_seg.data.copy_(batch["label_seg"])
        net.g.forward()
        net.det.join()
        multibox_metric.update(net)
        seg_prob = net.seg_out.prob.data                     # (B, h, w, ld) NHWC probabilities
        acc_metric.update([net.label_seg.data], [seg_prob])
        seg_metric.update([net.label_seg.data], [seg_prob])
        pred_det = filter_detections(net.det.out.data, score_thresh)
        det_metric.update([batch["label_det"][:, :, :5]], [pred_det[:, :, :6]])
        if full_res is not None:
            class_maps.append(prob_upsampling(seg_prob, full_res, len(seg_class_names)))
        if batch.get("disparity") is not None:
            depth_metric.update(batch["disparity"], list(pred_det[:, None]))
    out = {}
    names, values = multibox_metric.get()
    out.update(zip(names, values))
    name, value = acc_metric.get()
    out[name] = value
    for m in (det_metric, seg_metric, depth_metric):
        names, values = m.get()
        out.update(zip(names, values))
    if full_res is not None:
        out["class_maps"] = class_maps
    return out
