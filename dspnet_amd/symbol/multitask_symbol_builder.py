"""Multi-task training graph (counterpart of symbol/multitask_symbol_builder.py:442-593).

backbone -> SSD extras + multibox heads -> MultiBoxTarget -> SoftmaxOutput(cls) / smooth-L1(loc)
         -> MultiBoxDetection (monitoring output)
         -> segmentation decoder (pyramid pooling + bilinear sampling + 3x3 conv + 4x4/2 deconv)
            -> SoftmaxOutput(seg, grad_scale 4, ignore 255)

Outputs, in the reference's order (:592): cls_prob (B,C+1,N), loc_loss (B,N*5), cls_label (B,N),
det_out (B,N,7), seg_out (B,19,H/4,W/4).

Generalisations that are this build's own: the sampling grid is (H/8, W/8)
instead of the hard-coded (64,128) (equal at the reference's only consistent shape 512x1024), and the
(1,6) `affine_matrix` argument -- learnable, as in the reference (:574, multi_init.py:72) -- drives ONE
grid that is shared by every sample of the batch (MXNet's BilinearSampler wants one grid per sample, which
only binds at the reference's batch size 1); its gradient is summed over the batch.
"""
import torch

from .. import engine as E
from .. import functional as fn
from .. import operator as op
from . import resnet as resnet_mod
from . import vgg16_reduced as vgg_mod
from . import inceptionv3 as inception_mod
from .common import multi_layer_feature, multitask_layer

import os as _os
def _switch(name):
    """A/B switches of the side-stream schedule, read when a graph is BUILT: DSPN_TARGET_SIDE, DSPN_DET_SIDE, DSPN_DET_SIDE_BWD
    (all on by default; "0" puts that part back on the main stream -- results are bit-identical either way)"""
    return _os.environ.get(name, "1") != "0"
eps = 2e-5          # symbol/multitask_symbol_builder.py:5
seg_classes = 19    # :7


class MultiBoxTargetNode(E.Node):
    """mx.contrib.symbol.MultiBoxTarget(anchor, label, cls_preds, overlap_threshold=.5, ignore_label=-1,
    negative_mining_ratio=3, minimum_negative_samples=0, negative_mining_thresh=.5) (:517-521)"""

    def __init__(self, g, anchors, label, cls_flat, num_cls):
        self.anchors, self.label, self.cls_flat, self.C = anchors, label, cls_flat, num_cls
        B, N = cls_flat.shape[0], anchors.shape[1]
        self.cls_preds = g.tensor((B, num_cls, N), "multibox_cls_pred", requires_grad=False, dtype=torch.float32)
        cuda = g.device.type == "cuda"
        self.loc_target = torch.zeros(B, N * 5, device=g.device) if cuda else None
        self.loc_mask = torch.zeros(B, N * 5, device=g.device) if cuda else None
        self.cls_target = torch.zeros(B, N, device=g.device) if cuda else None
        # Round 4: the matching kernels are one workgroup per sample (0.3 ms with 7/8 of the CUs idle) and only the two
        # detection losses read their results, so they run on a second HIP stream beside whatever the graph builds between
        # this node and the losses (the segmentation decoder's forward, _build); the losses join() first
        self._g = g
        self.side = E.shared_stream(g.device, "target") if (cuda and _switch("DSPN_TARGET_SIDE")) else None
        self.ready = torch.cuda.Event() if self.side is not None else None
        self.done = torch.cuda.Event() if self.side is not None else None
        self.pending = False
        if self.side is not None:
            g.pre_forward.append(self.join)
        # the operator's temp space, owned by this node: it also holds the per-sample abort codes of THIS node's last
        # forward (two graphs on one device -- a training and an evaluation net -- never read each other's)
        self.ws = (torch.empty(op.target_workspace_bytes(B, N, label.shape[1]), dtype=torch.uint8, device=g.device)
                   if g.device.type == "cuda" else None)

    def _run(self):
        B, C, N = self.cls_preds.shape
        fn.transpose_bnc(self.cls_flat.data.view(B, N, C), out=self.cls_preds.data)
        out = None if self.loc_target is None else (self.loc_target, self.loc_mask, self.cls_target)
        self.loc_target, self.loc_mask, self.cls_target = op.MultiBoxTarget(
            self.anchors, self.label.data, self.cls_preds.data, overlap_threshold=.5, ignore_label=-1,
            negative_mining_ratio=3, minimum_negative_samples=0, negative_mining_thresh=.5,
            variances=(0.1, 0.1, 0.2, 0.2), workspace=self.ws, out=out)

    def forward(self):
        if self.side is None or self._g.side_segment is not None:     # (the whole detection branch is on a side stream already)
            self._run()
            return
        main = torch.cuda.current_stream(self.cls_preds.data.device)
        self.ready.record(main)
        self.side.wait_event(self.ready)
        with torch.cuda.stream(self.side):
            self._run()
            self.done.record(self.side)
        self.pending = True

    def join(self):
        """order the current stream behind the matching kernels (the losses call it; idempotent)"""
        self._g.join_side()
        if self.pending:
            torch.cuda.current_stream(self.cls_preds.data.device).wait_event(self.done)
            self.pending = False

    def raise_on_errors(self):
        """the reference's data-dependent CHECKs (multibox_target.cc:98-101, :236) for the last forward: synchronises;
        the kernel itself only records the codes so that the step never waits on the host"""
        self.join()
        op.MultiBoxTarget_check(self.cls_preds.shape[0], self.cls_preds.data.device, workspace=self.ws)


class ClsSoftmaxOutput(E.Node):
    """SoftmaxOutput(cls_preds, cls_target, ignore_label=-1, use_ignore, multi_output,
    normalization='valid', grad_scale=1) (:526-528)"""

    def __init__(self, g, cls_flat, target_node, num_cls):
        self.x, self.tn, self.C = cls_flat, target_node, num_cls
        B = cls_flat.shape[0]
        N = cls_flat.shape[1] // num_cls
        self.prob_nc = fn.zeros(B, N, num_cls, device=g.device)
        self.gbuf = fn.zeros(*cls_flat.shape, device=g.device)   # d loss / d logits, made in forward
        self.valid = fn.zeros(1, device=g.device)
        self.cls_prob = g.tensor((B, num_cls, N), "cls_prob", requires_grad=False, dtype=torch.float32)

    def forward(self):
        B, C, N = self.cls_prob.shape
        self.tn.join()
        fn.count(self.tn.cls_target, "ne", -1.0, out=self.valid)
        fn.softmax_output(self.x.data.view(B * N, C), self.tn.cls_target, C, -1.0, 1.0, self.valid,
                          prob=self.prob_nc.view(B * N, C), grad=self.gbuf.view(B * N, C))
        fn.transpose_bnc(self.prob_nc, out=self.cls_prob.data)

    def backward(self):
        self.x.give_grad(self.gbuf)


class LocLoss(E.Node):
    """MakeLoss(smooth_l1(loc_target_mask * (loc_preds - loc_target), scalar=1), normalization='valid')
    (:529-532)"""

    def __init__(self, g, loc_preds, target_node):
        self.x, self.tn = loc_preds, target_node
        self.valid = fn.zeros(1, device=g.device)
        self.out = g.tensor(loc_preds.shape, "loc_loss", requires_grad=False, dtype=torch.float32)

    def forward(self):
        self.tn.join()
        fn.smooth_l1_forward(self.x.data, self.tn.loc_target, self.tn.loc_mask, out=self.out.data)
        fn.count(self.out.data, "gt", 0.0, out=self.valid)

    def backward(self):
        if self.x._gw:
            tmp = fn.smooth_l1_backward(self.x.data, self.tn.loc_target, self.tn.loc_mask, self.valid, 1.0)
            fn.add(self.x.grad, tmp, out=self.x.grad)
        else:
            dx, _ = self.x.grad_target()
            fn.smooth_l1_backward(self.x.data, self.tn.loc_target, self.tn.loc_mask, self.valid, 1.0, out=dx)


class Detection(E.Node):
    """MultiBoxDetection(cls_prob, loc_preds, anchors, nms_threshold, force_suppress, nms_topk) wrapped in
    a zero-gradient MakeLoss (:536-539)"""

    def __init__(self, g, cls_prob, loc_preds, anchors, nms_thresh, force_suppress, nms_topk):
        self.cls_prob, self.loc_preds, self.anchors = cls_prob, loc_preds, anchors
        self.kw = dict(nms_threshold=nms_thresh, force_suppress=force_suppress,
                       variances=(0.1, 0.1, 0.2, 0.2), nms_topk=nms_topk)
        B, N = cls_prob.shape[0], anchors.shape[1]
        self.out = g.tensor((B, N, 7), "det_out", requires_grad=False, dtype=torch.float32)
        # det_out is a monitoring output: nothing in the step consumes it and it only reads cls_prob / loc_preds,
        # so its one-workgroup-per-sample sort + NMS kernels run on a second HIP stream beside the decoder and the
        # backward pass instead of leaving 7/8 of the CUs idle for ~1.3 ms.  join() (called at the top of the next
        # forward, before cls_prob / loc_preds are overwritten, and by outputs()) orders the main stream behind it.
        self.side = E.shared_stream(g.device, "detection") if g.device.type == "cuda" else None
        self.ready = torch.cuda.Event() if self.side is not None else None
        self.done = torch.cuda.Event() if self.side is not None else None
        self.pending = False
        self._g = g
        g.pre_forward.append(self.join)
        # the operator's temp space, owned by this node (as MultiBoxTargetNode's): its kernels run on a side stream, beside
        # whatever another graph or an eager caller on this device does with the shared buffer
        if g.device.type == "cuda":
            self.kw["workspace"] = torch.empty(op.detection_workspace_bytes(B, N), dtype=torch.uint8, device=g.device)

    def forward(self):
        if self.side is None:
            op.MultiBoxDetection(self.cls_prob.data, self.loc_preds.data, self.anchors, out=self.out.data, **self.kw)
            return
        main = torch.cuda.current_stream(self.out.data.device)
        self.ready.record(main)
        self.side.wait_event(self.ready)
        with torch.cuda.stream(self.side):
            op.MultiBoxDetection(self.cls_prob.data, self.loc_preds.data, self.anchors, out=self.out.data, **self.kw)
            self.done.record(self.side)
        self.pending = True

    def join(self):
        self._g.join_side()        # (test graphs: the whole detection branch, this node included, runs on the branch stream)
        if self.pending:
            torch.cuda.current_stream(self.out.data.device).wait_event(self.done)
            self.pending = False


class SegSoftmaxOutput(E.Node):
    """SoftmaxOutput(score4_conv, multi_output, grad_scale=4, use_ignore, ignore_label=255) (:588).
    normalization is the default 'null'; MXNet then divides the gradient by the number of spatial
    positions of one sample (softmax_output-inl.h, multi_output branch)."""

    def __init__(self, g, logits, label, classes):
        self.x, self.label, self.C = logits, label, classes
        B, H, W, Cp = logits.shape
        self.prob = g.tensor(logits.shape, "seg_prob_nhwc", requires_grad=False, dtype=torch.float32)
        self.gbuf = fn.zeros(*logits.shape, device=g.device, dtype=logits.dtype)     # a gradient of an activation
        self.scale = 4.0 / float(H * W)

    def forward(self):
        B, H, W, Cp = self.x.shape
        fn.softmax_output(self.x.data.view(B * H * W, Cp), self.label.data, self.C, 255.0, self.scale, None,
                          prob=self.prob.data.view(B * H * W, Cp), grad=self.gbuf.view(B * H * W, Cp))

    def backward(self):
        self.x.give_grad(self.gbuf)

    def nchw(self):
        """seg_out in the reference's layout (B, 19, H/4, W/4)"""
        return fn.nhwc_to_nchw(self.prob.data, self.C)


class ClsSoftmaxActivation(E.Node):
    """SoftmaxActivation(cls_preds, mode='channel') of the test graph (:667-668): probabilities only"""

    def __init__(self, g, cls_flat, num_cls):
        self.x, self.C = cls_flat, num_cls
        B = cls_flat.shape[0]
        N = cls_flat.shape[1] // num_cls
        self.prob_nc = fn.zeros(B, N, num_cls, device=g.device)
        self.cls_prob = g.tensor((B, num_cls, N), "cls_prob", requires_grad=False, dtype=torch.float32)

    def forward(self):
        B, C, N = self.cls_prob.shape
        fn.softmax_output(self.x.data.view(B * N, C), None, C, -1.0, 1.0, None,
                          prob=self.prob_nc.view(B * N, C), want_grad=False)
        fn.transpose_bnc(self.prob_nc, out=self.cls_prob.data)


class SegSoftmax(E.Node):
    """mx.symbol.softmax over the class axis of score4_conv (:723): probabilities only"""

    def __init__(self, g, logits, classes):
        self.x, self.C = logits, classes
        self.prob = g.tensor(logits.shape, "seg_prob_nhwc", requires_grad=False, dtype=torch.float32)

    def forward(self):
        B, H, W, Cp = self.x.shape
        fn.softmax_output(self.x.data.view(B * H * W, Cp), None, self.C, 255.0, 1.0, None,
                          prob=self.prob.data.view(B * H * W, Cp), want_grad=False)

    def nchw(self):
        return fn.nhwc_to_nchw(self.prob.data, self.C)


class MultiTaskNet:
    """what symbol.bind(...) returns in the reference: inputs, executor graph, outputs"""

    def __init__(self, g, data, label_det, label_seg, nodes):
        self.g, self.data, self.label_det, self.label_seg = g, data, label_det, label_seg
        self.__dict__.update(nodes)

    def outputs(self):
        """training graph: [cls_prob, loc_loss, cls_label, det_out, seg_out] (multitask_symbol_builder.py:592);
        test graph: [det, seg_out] (:726); segmentation-only graphs: [seg_out] (:322, :439); detection-only
        test graph: [det] (:208)"""
        if self.det is None:
            return [self.seg_out.nchw()]
        self.det.join()
        if self.target is None:
            return [self.det.out.data] if self.seg_out is None else [self.det.out.data, self.seg_out.nchw()]
        outs = [self.cls_out.cls_prob.data, self.loc_loss.out.data, self.target.cls_target, self.det.out.data]
        return outs if self.seg_out is None else outs + [self.seg_out.nchw()]


def get_multi_symbol_train(network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                           normalizations=-1, steps=(), min_filter=128, nms_thresh=0.5, force_suppress=False,
                           nms_topk=400, batch_size=1, data_shape=(3, 512, 1024), num_labels=200, device=None,
                           num_layers=50, seed=0, **kwargs):
    """symbol/multitask_symbol_builder.py:442-593"""
    return _build(True, True, network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios, normalizations,
                  steps, min_filter, nms_thresh, force_suppress, nms_topk, batch_size, data_shape, num_labels, device,
                  num_layers, seed)


def get_multi_symbol(network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                     normalizations=-1, steps=(), min_filter=128, nms_thresh=0.5, force_suppress=False,
                     nms_topk=400, batch_size=1, data_shape=(3, 512, 1024), device=None, num_layers=50, seed=0,
                     **kwargs):
    """Test graph, symbol/multitask_symbol_builder.py:595-726: outputs [det, seg_out].  No label inputs, no
    MultiBoxTarget, no losses; class probabilities by SoftmaxActivation(mode='channel').  BatchNorm still
    uses batch statistics, as every shipped caller runs with is_train=True (detect/multitask_detector.py:228).
    The reference's `mx.symbol.softmax(..., multi_output=True)` on the seg logits is read as a softmax over
    the class axis."""
    return _build(False, True, network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                  normalizations, steps, min_filter, nms_thresh, force_suppress, nms_topk, batch_size, data_shape,
                  200, device, num_layers, seed)


def get_det_symbol_train(network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                         normalizations=-1, steps=(), min_filter=128, nms_thresh=0.5, force_suppress=False,
                         nms_topk=400, batch_size=1, data_shape=(3, 300, 300), num_labels=200, device=None,
                         num_layers=50, seed=0, **kwargs):
    """Detection + depth only, symbol/multitask_symbol_builder.py:20-121: outputs [cls_prob, loc_loss, cls_label,
    det_out]; the same graph without the segmentation decoder."""
    return _build(True, False, network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                  normalizations, steps, min_filter, nms_thresh, force_suppress, nms_topk, batch_size, data_shape,
                  num_labels, device, num_layers, seed)


def get_det_symbol(network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                   normalizations=-1, steps=(), min_filter=128, nms_thresh=0.5, force_suppress=False,
                   nms_topk=400, batch_size=1, data_shape=(3, 300, 300), device=None, num_layers=50, seed=0,
                   **kwargs):
    """Detection + depth test graph, symbol/multitask_symbol_builder.py:123-209: output [det]"""
    return _build(False, False, network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                  normalizations, steps, min_filter, nms_thresh, force_suppress, nms_topk, batch_size, data_shape,
                  200, device, num_layers, seed)


def get_seg_symbol_train(network, num_classes, from_layers, num_filters=None, strides=None, pads=None, sizes=None,
                         ratios=None, normalizations=-1, steps=(), min_filter=128, nms_thresh=0.5,
                         force_suppress=False, nms_topk=400, batch_size=1, data_shape=(3, 512, 1024), device=None,
                         num_layers=50, seed=0, **kwargs):
    """Segmentation only, symbol/multitask_symbol_builder.py:211-323: backbone -> pyramid decoder ->
    SoftmaxOutput(grad_scale=4, ignore 255); output [seg_out].  The SSD arguments are accepted and unused, as
    in the reference.  Gradient reaches the backbone through conv_feat only (res3 / res4 are BlockGrad'ed)."""
    return _build(True, True, network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                  normalizations, steps, min_filter, nms_thresh, force_suppress, nms_topk, batch_size, data_shape,
                  0, device, num_layers, seed, with_det=False)


def get_seg_symbol(network, num_classes, from_layers, num_filters=None, strides=None, pads=None, sizes=None,
                   ratios=None, normalizations=-1, steps=(), min_filter=128, nms_thresh=0.5, force_suppress=False,
                   nms_topk=400, batch_size=1, data_shape=(3, 512, 1024), device=None, num_layers=50, seed=0,
                   **kwargs):
    """Segmentation test graph, symbol/multitask_symbol_builder.py:325-440: output [seg_out] (softmax over classes)"""
    return _build(False, True, network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                  normalizations, steps, min_filter, nms_thresh, force_suppress, nms_topk, batch_size, data_shape,
                  0, device, num_layers, seed, with_det=False)


def _detection_branch(g, train, internals, label, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
                      normalizations, steps, min_filter, nms_thresh, force_suppress, nms_topk):
    """SSD feature layers, heads, target matching / losses (training) and MultiBoxDetection (:502-539)"""
    # remove res3 from the input layers of SSD (:502-508).  The reference slices from_layers / num_filters /
    # strides / pads / sizes / ratios but not `normalizations` and `steps` (a list-valued preset then trips
    # the length asserts of symbol/common.py:350,356); this build slices those too.
    from_layers, num_filters, strides, pads = from_layers[1:], num_filters[1:], strides[1:], pads[1:]
    sizes, ratios = sizes[1:], ratios[1:]
    if isinstance(normalizations, (list, tuple)):
        normalizations = list(normalizations)[1:]
    steps = list(steps)[1:] if steps else steps

    layers = multi_layer_feature(g, internals, from_layers, num_filters, strides, pads, min_filter=min_filter)
    # conv_feat: the reference reads internals[from_layers[2]] (:500), which only exists for the resnet-50
    # preset; presets whose third entry is '' (an extra layer) use that extra layer -- the stride-32 map --
    # in the same role (build's own wiring)
    conv_feat = layers[1]
    loc_preds, cls_flat, anchor_boxes = multitask_layer(g, layers, num_classes, sizes=sizes, ratios=ratios,
                                                        normalization=normalizations, clip=False, steps=steps)
    ncls = num_classes + 1
    if train:
        target = g.add(MultiBoxTargetNode(g, anchor_boxes, label, cls_flat, ncls))
        cls_out = g.add(ClsSoftmaxOutput(g, cls_flat, target, ncls))
        loc_loss = g.add(LocLoss(g, loc_preds, target))
    else:
        target, loc_loss = None, None
        cls_out = g.add(ClsSoftmaxActivation(g, cls_flat, ncls))
    det = g.add(Detection(g, cls_out.cls_prob, loc_preds, anchor_boxes, nms_thresh, force_suppress, nms_topk))

    return conv_feat, target, cls_out, loc_loss, det, anchor_boxes, loc_preds, cls_flat


def _build(train, with_seg, network, num_classes, from_layers, num_filters, strides, pads, sizes, ratios, normalizations,
           steps, min_filter, nms_thresh, force_suppress, nms_topk, batch_size, data_shape, num_labels, device,
           num_layers, seed, with_det=True):
    assert network in ("resnet", "vgg16_reduced", "inceptionv3"), "backbones: resnet, vgg16_reduced, inceptionv3"
    device = device or torch.device("cuda", torch.cuda.current_device())
    g = E.Graph(device)
    C, H, W = data_shape
    data = g.tensor((batch_size, C, H, W), "data", requires_grad=False, dtype=torch.float32)
    label = (g.tensor((batch_size, num_labels, 6), "label_det", requires_grad=False, dtype=torch.float32)
             if (train and with_det) else None)
    seg_label = (g.tensor((batch_size, H // 4, W // 4), "seg_out_label", requires_grad=False, dtype=torch.float32)
                 if (train and with_seg) else None)

    if network == "resnet":
        internals = resnet_mod.get_symbol(g, data, num_layers=num_layers)
    elif network == "inceptionv3":
        internals = inception_mod.get_symbol(g, data)
    else:
        internals = vgg_mod.get_symbol(g, data)
    res3 = internals[from_layers[0] + "_output"]
    res4 = internals[from_layers[1] + "_output"]

    det_first = len(g.nodes) if with_det else None       # the first node of the detection branch
    if with_det:
        (conv_feat, target, cls_out, loc_loss, det, anchor_boxes, loc_preds, cls_flat) = _detection_branch(
            g, train, internals, label, num_classes, from_layers, num_filters, strides, pads, sizes, ratios,
            normalizations, steps, min_filter, nms_thresh, force_suppress, nms_topk)
    else:
        # segmentation-only graphs read the third backbone map directly (:269, :383): no SSD extra layers exist
        assert from_layers[2].strip(), \
            "segmentation-only graphs need a backbone layer as from_layers[2] (the resnet presets)"
        conv_feat = internals[from_layers[2].strip() + "_output"]
        target = cls_out = loc_loss = det = anchor_boxes = loc_preds = cls_flat = None

    if not with_seg:
        g.finalize(seed)
        return MultiTaskNet(g, data, label, None,
                            dict(target=target, cls_out=cls_out, loc_loss=loc_loss, det=det, seg_out=None,
                                 anchors=anchor_boxes, loc_preds=loc_preds, cls_flat=cls_flat))

    # segmentation task (pyramid pooling module) (:541-589)
    def conv_bn(x, name, nf, k, pad, tap_expand=False):
        c = g.add(E.Conv(g, x, name, nf, k, 1, pad, init="maxdim", tap_expand=tap_expand)).out
        return g.add(E.BatchNorm(g, c, name + "_bn", fix_gamma=True, eps=eps)).out

    res3_block = g.add(E.BlockGrad(g, res3, "res3_block")).out
    res3_reduced_bn = conv_bn(res3_block, "res3_reduced", 128, 1, 0)
    res3_reduced2_bn = conv_bn(res3_reduced_bn, "res3_reduced2", 128, 3, 1)
    res4_block = g.add(E.BlockGrad(g, res4, "res4_block")).out
    res4_reduced_bn = conv_bn(res4_block, "res4_reduced", 256, 1, 0)
    res4_reduced2_bn = conv_bn(res4_reduced_bn, "res4_reduced2", 256, 3, 1)
    # the reference also declares a 1x1 "res5_reduced" conv whose output is never used (:556-558)
    res5_reduced_bn = g.add(E.BatchNorm(g, conv_feat, "res5_reduced_bn", fix_gamma=True, eps=eps)).out
    score_pool1 = g.add(E.AvgPool(g, res5_reduced_bn, "score_pool1", 1)).out
    score_pool2 = g.add(E.AvgPool(g, res5_reduced_bn, "score_pool2", 2)).out
    score_pool4 = g.add(E.AvgPool(g, res5_reduced_bn, "score_pool4", 4)).out
    score2_pool4_bn = conv_bn(score_pool4, "score2_pool4", 128, 1, 0)
    score2_pool2_bn = conv_bn(score_pool2, "score2_pool2", 256, 1, 0)
    score2_pool1_bn = conv_bn(score_pool1, "score2_pool1", 512, 1, 0)
    target_hw = (H // 8, W // 8)   # (64,128) at 512x1024 (:575)
    affine_matrix = E.affine_matrix_param(g)        # mx.sym.var("affine_matrix", shape=(1,6)) (:574)
    pyramid = [score2_pool4_bn, score2_pool2_bn, score2_pool1_bn, res5_reduced_bn, res4_reduced2_bn, res3_reduced2_bn]
    if E.COMMUTE_RESIZE_CONV:
        # score3_conv over score3_concat without the 3328-channel concatenation (engine.BilinearConcatConv)
        c = g.add(E.BilinearConcatConv(g, pyramid, "score3_conv", seg_classes, 3, 1, target_hw, affine_matrix)).out
        score3_conv_bn = g.add(E.BatchNorm(g, c, "score3_conv_bn", fix_gamma=True, eps=eps)).out
    else:
        score3_concat = g.add(E.BilinearConcat(g, pyramid, "score3_concat", target_hw, affine_matrix)).out
        score3_conv_bn = conv_bn(score3_concat, "score3_conv", seg_classes, 3, 1, tap_expand=True)
    score4_conv = g.add(E.Deconv4x4s2(g, score3_conv_bn, "score4_conv", seg_classes)).out
    if train:
        seg_out = g.add(SegSoftmaxOutput(g, score4_conv, seg_label, seg_classes))
    else:
        seg_out = g.add(SegSoftmax(g, score4_conv, seg_classes))

    if train and with_det and target is not None and _switch("DSPN_TARGET_SIDE"):
        # the detection losses (and MultiBoxDetection behind them) run AFTER the segmentation decoder's forward: the target
        # matching kernels, issued on their own stream when the heads are done, have the decoder's ~1.5 ms to finish in.
        # Backward order among the writers of a shared gradient is unchanged (decoder before heads, as before).
        late = [cls_out, loc_loss, det]
        for n in late:
            g.nodes.remove(n)
        at = g.nodes.index(seg_out)
        g.nodes[at:at] = late
        if _switch("DSPN_DET_SIDE") and det_first is not None:
            # ... and the whole detection branch in front of them -- extra layers, heads, packing, matching -- runs its FORWARD
            # on that stream, from the first node the branch added (backward stays on the main stream: it accumulates into
            # gradients the decoder also writes)
            g.set_side_segment(det_first, g.nodes.index(target))
            if _switch("DSPN_DET_SIDE_BWD"):
                # ... and, in backward, the part of the branch whose gradients stay inside it -- head packing, the heads on the
                # extra maps, the extra layers behind the first -- beside the decoder's backward; the heads on backbone maps
                # and the first extra layer write gradients the decoder also writes and stay on the main stream
                from .common import HeadPack
                hi = g.nodes.index(target)
                side = [i for i in range(det_first + 1, hi + 1)
                        if isinstance(g.nodes[i], HeadPack)
                        or (isinstance(g.nodes[i], E.Conv) and g.nodes[i].w.name.startswith("multi_feat_"))]
                g.set_side_backward(side, det_first, hi, g.nodes.index(cls_out))
    if (not train) and with_det and det is not None and _switch("DSPN_DET_SIDE"):
        # test graph: the detection branch up to and including MultiBoxDetection beside the segmentation decoder; its only
        # reader is the caller, through det.join()
        g.set_side_segment(det_first, g.nodes.index(det))
    g.finalize(seed)
    return MultiTaskNet(g, data, label, seg_label,
                        dict(target=target, cls_out=cls_out, loc_loss=loc_loss, det=det, seg_out=seg_out,
                             anchors=anchor_boxes, loc_preds=loc_preds, cls_flat=cls_flat))
