"""MXNet-contrib-compatible front end of the three SSD operators.

Mirrors `mx.contrib.symbol.MultiBoxPrior / MultiBoxTarget / MultiBoxDetection`
as this repo's reference registers them (operator/multibox_prior.cc:96,
multibox_target.cc:308, multibox_detection.cc:194): same names, keyword
arguments, defaults, output shapes and shape-check error texts.  Tensors are
torch CUDA tensors used purely as device buffers; all arithmetic happens in the
HIP kernels behind the C ABI of include/dspn_multibox.h.

The outputs carry no gradient (the reference's Backward writes zeros).
"""
import ctypes

import torch

from . import _lib
from ._lib import DspnError, check

_ws_cache = {}


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(nbytes, device, tag):
    key = (tag, device)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _dev_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise DspnError(f"{name}: must live on the GPU (there is no CPU path)")
    if t.dtype != torch.float32:
        raise DspnError(f"{name}: only float32 is supported")
    return t.contiguous()  # reference CHECKs contiguity (multibox_target-inl.h:125-131)


def _tuple(v):
    if isinstance(v, str):  # MXNet accepts "(0.1,0.2)" strings (symbol/common.py:386-390)
        v = v.strip().strip("()[]")
        return tuple(float(x) for x in v.split(",") if x.strip())
    if isinstance(v, (int, float)):
        return (float(v),)
    return tuple(float(x) for x in v)


def MultiBoxPrior(data, sizes=(1.0,), ratios=(1.0,), clip=False, steps=(-1.0, -1.0),
                  offsets=(0.5, 0.5), out=None):
    """Anchor boxes for one feature map.  data: (B, C, H, W)-shaped tensor (only
    H, W are read) or an (H, W) tuple.  Returns (1, H*W*(len(sizes)+len(ratios)-1), 4).
    Shape rules: multibox_prior-inl.h:171-193."""
    sizes, ratios, steps, offsets = _tuple(sizes), _tuple(ratios), _tuple(steps), _tuple(offsets)
    if isinstance(data, torch.Tensor):
        if data.dim() < 4:
            raise DspnError("Input data should be 4D: batch-channel-y-x")
        h, w, device = int(data.shape[2]), int(data.shape[3]), data.device
    else:
        h, w = int(data[0]), int(data[1])
        device = torch.device("cuda", torch.cuda.current_device())
    if len(steps) != 2:
        raise DspnError("Step ndim must be 2: (step_y, step_x)")
    if len(offsets) != 2:
        raise DspnError("MultiBoxPrior: offsets must have 2 values")
    if len(sizes) == 0 or len(ratios) == 0:
        raise DspnError("MultiBoxPrior: sizes and ratios must not be empty")
    n = h * w * (len(sizes) + len(ratios) - 1)
    if out is None:
        out = torch.empty((1, max(n, 0), 4), dtype=torch.float32, device=device)
    L = _lib.lib()
    check(L.dspn_multibox_prior_f32(_lib.floats(sizes), len(sizes), _lib.floats(ratios),
                                    len(ratios), h, w, steps[0], steps[1], offsets[0],
                                    offsets[1], int(bool(clip)),
                                    ctypes.c_void_p(out.data_ptr()), _stream_ptr()),
          "MultiBoxPrior")
    return out


def MultiBoxTarget(anchor, label, cls_pred, overlap_threshold=0.5, ignore_label=-1.0,
                   negative_mining_ratio=-1.0, negative_mining_thresh=0.5,
                   minimum_negative_samples=0, variances=(0.1, 0.1, 0.2, 0.2),
                   check_errors=False, workspace=None, out=None):
    """Training targets.  anchor (1,N,4), label (B,L,6), cls_pred (B,C+1,N) ->
    [loc_target (B,N*5), loc_mask (B,N*5), cls_target (B,N)].
    out: optional (loc_target, loc_mask, cls_target) float32 device buffers of those shapes to write into.
    Shape rules and messages: multibox_target-inl.h:213-238.
    check_errors=True synchronises and raises on the data-dependent aborts of the
    reference (multibox_target.cc:98-101, :236).
    workspace: a caller-owned uint8 device buffer of at least target_workspace_bytes(B, N, L) bytes (the reference's
    kTempSpace request); it also receives the per-sample abort codes MultiBoxTarget_check reads.  Default: a buffer
    shared by every call on the device."""
    variances = _tuple(variances)
    if anchor.dim() != 3:
        raise DspnError("Anchor should be batch shared N*4 tensor")
    if anchor.shape[0] != 1:
        raise DspnError("Anchors are shared across batches, first dim=1")
    if anchor.shape[1] <= 0:
        raise DspnError("Number boxes should > 0")
    if anchor.shape[2] != 4:
        raise DspnError("Box dimension should be 4: [xmin-ymin-xmax-ymax]")
    if label.dim() != 3:
        raise DspnError("Label should be [batch-num_labels-(>=5)] tensor")
    if label.shape[1] <= 0:
        raise DspnError("Padded label should > 0")
    if label.shape[2] != 6:
        raise DspnError("Label width should be 6: [cls-xmin-ymin-xmax-ymax-dist]")
    if cls_pred.dim() != 3:
        raise DspnError("Prediction: [nbatch-num_classes-num_anchors]")
    if cls_pred.shape[2] != anchor.shape[1]:
        raise DspnError("Number of anchors mismatch")
    if len(variances) != 4:
        raise DspnError("MultiBoxTarget: variances must have 4 values")
    anchor, label, cls_pred = (_dev_f32(anchor, "anchor"), _dev_f32(label, "label"),
                               _dev_f32(cls_pred, "cls_pred"))
    B, N, Lr = int(label.shape[0]), int(anchor.shape[1]), int(label.shape[1])
    dev = anchor.device
    if out is not None:
        loc_target, loc_mask, cls_target = out
        for t, shp in ((loc_target, (B, N * 5)), (loc_mask, (B, N * 5)), (cls_target, (B, N))):
            if tuple(t.shape) != shp or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
                raise DspnError(f"MultiBoxTarget: out buffers must be contiguous float32 {shp} tensors on {dev}")
    else:
        loc_target = torch.empty((B, N * 5), dtype=torch.float32, device=dev)
        loc_mask = torch.empty((B, N * 5), dtype=torch.float32, device=dev)
        cls_target = torch.empty((B, N), dtype=torch.float32, device=dev)
    L = _lib.lib()
    nbytes = L.dspn_multibox_target_workspace_bytes(B, N, Lr)
    if workspace is not None:
        if workspace.dtype != torch.uint8 or workspace.device != dev or workspace.numel() < nbytes:
            raise DspnError(f"MultiBoxTarget: workspace must be a uint8 tensor of >= {nbytes} bytes on {dev}")
        ws = workspace
    else:
        ws = _workspace(nbytes, dev, "target")
    check(L.dspn_multibox_target_f32(
        anchor.data_ptr(), label.data_ptr(), cls_pred.data_ptr(), B, N, Lr, int(label.shape[2]),
        int(cls_pred.shape[1]), float(overlap_threshold), float(ignore_label),
        float(negative_mining_ratio), float(negative_mining_thresh),
        int(minimum_negative_samples), _lib.floats(variances), loc_target.data_ptr(),
        loc_mask.data_ptr(), cls_target.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr()),
        "MultiBoxTarget")
    if check_errors:
        codes = (ctypes.c_int * B)()
        check(L.dspn_multibox_target_errors(ws.data_ptr(), B, codes, _stream_ptr()),
              "MultiBoxTarget")
    return [loc_target, loc_mask, cls_target]


def target_workspace_bytes(batch, num_anchors, num_labels):
    return int(_lib.lib().dspn_multibox_target_workspace_bytes(int(batch), int(num_anchors), int(num_labels)))


def detection_workspace_bytes(batch, num_anchors):
    return int(_lib.lib().dspn_multibox_detection_workspace_bytes(int(batch), int(num_anchors)))


def MultiBoxTarget_check(batch, device, workspace=None):
    """Deferred form of check_errors=True: reads the per-sample abort codes a MultiBoxTarget call of `batch` samples left
    in its workspace (synchronises) and raises DspnError with the reference's message if a label row after the
    terminator was not all -1 (multibox_target.cc:98-101) or hard-negative mining ran out of candidates (:236).  A
    training loop calls this once per step where it synchronises anyway (solver.fit does, at the metric read-out).
    workspace: the buffer that call was given (a graph node keeps its own, so two graphs on one device never read each
    other's codes); default: the shared buffer, i.e. the LAST call on `device` without a workspace of its own."""
    ws = workspace if workspace is not None else _ws_cache.get(("target", device))
    if ws is None:
        return
    L = _lib.lib()
    codes = (ctypes.c_int * int(batch))()
    check(L.dspn_multibox_target_errors(ws.data_ptr(), int(batch), codes, _stream_ptr()), "MultiBoxTarget")


def MultiBoxDetection(cls_prob, loc_pred, anchor, clip=True, threshold=0.01, background_id=0,
                      nms_threshold=0.5, force_suppress=False, variances=(0.1, 0.1, 0.2, 0.2),
                      nms_topk=-1, out=None, workspace=None):
    """Decode + NMS.  cls_prob (B,C+1,N), loc_pred (B,N*5), anchor (1,N,4) ->
    (B,N,7) rows [id, score, xmin, ymin, xmax, ymax, dist], id=-1 for empty rows.
    Shape rules: multibox_detection-inl.h:149-171.  background_id is accepted and,
    as in the reference kernels, class 0 is always the background.
    workspace: a caller-owned uint8 buffer of detection_workspace_bytes(B, N) (the operator's kTempSpace,
    multibox_detection-inl.h:183-186).  A graph node keeps its own, so two graphs -- or one graph's side stream and an
    eager caller -- never share scratch; default: one buffer per device, for callers on ONE stream."""
    variances = _tuple(variances)
    if cls_prob.dim() != 3:
        raise DspnError(f"Provided: {tuple(cls_prob.shape)}")
    if loc_pred.dim() != 2:
        raise DspnError(f"Provided: {tuple(loc_pred.shape)}")
    if anchor.dim() != 3:
        raise DspnError(f"Provided: {tuple(anchor.shape)}")
    if cls_prob.shape[2] != anchor.shape[1]:
        raise DspnError("Number of anchors mismatch")
    if cls_prob.shape[2] * 5 != loc_pred.shape[1]:
        raise DspnError("# anchors mismatch with # loc")
    if anchor.shape[1] <= 0:
        raise DspnError("Number of anchors must > 0")
    if anchor.shape[2] != 4:
        raise DspnError("Box dimension should be 4: [xmin-ymin-xmax-ymax]")
    if len(variances) != 4:
        raise DspnError("Variance size must be 4")
    cls_prob, loc_pred, anchor = (_dev_f32(cls_prob, "cls_prob"), _dev_f32(loc_pred, "loc_pred"),
                                  _dev_f32(anchor, "anchor"))
    B, N = int(cls_prob.shape[0]), int(anchor.shape[1])
    dev = cls_prob.device
    if out is None:
        out = torch.empty((B, N, 7), dtype=torch.float32, device=dev)
    L = _lib.lib()
    nbytes = L.dspn_multibox_detection_workspace_bytes(B, N)
    if workspace is not None:
        if workspace.numel() < nbytes or not workspace.is_cuda or workspace.dtype != torch.uint8:
            raise DspnError(f"MultiBoxDetection: workspace needs {nbytes} bytes of uint8 device memory")
        ws = workspace
    else:
        ws = _workspace(nbytes, dev, "detection")
    check(L.dspn_multibox_detection_f32(
        cls_prob.data_ptr(), loc_pred.data_ptr(), anchor.data_ptr(), B, N,
        int(cls_prob.shape[1]), float(threshold), int(bool(clip)), _lib.floats(variances),
        float(nms_threshold), int(bool(force_suppress)), int(nms_topk), out.data_ptr(),
        ws.data_ptr(), ws.numel(), _stream_ptr()), "MultiBoxDetection")
    return out
