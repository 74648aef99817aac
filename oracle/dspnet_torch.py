"""CPU restatement of the DSPNet multi-task training graph in plain PyTorch ops.
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

PARITY STATUS: "parity unpinned".  The reference graph (symbol/multitask_symbol_builder.py:442-593
on symbol/resnet.py, symbol/common.py) is a set of MXNet symbols; MXNet is neither vendored nor
installable here and the reference has no tests or recorded outputs beyond tensor SHAPES
(utils.py:38), so the arithmetic of every built-in operator below is this file's reading of the
documented MXNet 0.11-1.0 semantics:
  * Convolution / Deconvolution(no bias) / Pooling(max: pad ignored, avg) -> torch conv2d /
    conv_transpose2d / max_pool2d / avg_pool2d;
  * BatchNorm(is_train): biased batch variance, eps 2e-5, fix_gamma => gamma == 1;
  * GridGenerator(transform_type='affine') on the learnable (1,6) `affine_matrix` + BilinearSampler ->
    affine_grid(align_corners=True) + grid_sample(bilinear, zeros padding, align_corners=True); one grid
    for the whole batch (the reference binds at batch size 1); autograd supplies d/d affine_matrix;
  * SoftmaxOutput(multi_output, use_ignore): backward (p - onehot) * grad_scale, zero on ignored
    labels, divided by #valid for normalization='valid', by the number of positions per sample for
    the default normalization; expressed here as the scalar whose autograd gradient is that;
  * MakeLoss(smooth_l1, normalization='valid'): gradient 1 / max(1, #(loss > 0)).
The SSD operators inside the graph call oracle/multibox_oracle.c.

The same function doubles as the CPU baseline ("port") timed by bench.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import multibox as om

EPS = 2e-5

# conv_quant="bf16" (forward_loss): every convolution GEMM -- forward, data gradient, weight gradient, and the 4x4/2
# transposed convolution likewise -- sees its two operands rounded to bfloat16 (round to nearest even) and accumulates
# exactly: the arithmetic of BASELINE.json configs[3] "bf16 MFMA convs" (bf16 operands, fp32 accumulate) with the
# accumulation error removed.  Everything else (BatchNorm, pooling, sampler, losses) stays in `dtype`.
_QUANT = None


def _q(t):
    return t.to(torch.bfloat16).to(t.dtype) if _QUANT == "bf16" else t


class _QuantGemmOp(torch.autograd.Function):
    """y = op(q(x), q(w)); backward: the two gradient GEMMs of the same op on q(dy) and the saved q(x) / q(w)"""

    @staticmethod
    def forward(ctx, x, w, op):
        xq, wq = _q(x.detach()), _q(w.detach())
        ctx.save_for_backward(xq, wq)
        ctx.op = op
        return op(xq, wq)

    @staticmethod
    def backward(ctx, gy):
        xq, wq = ctx.saved_tensors
        with torch.enable_grad():
            xr, wr = xq.clone().requires_grad_(), wq.clone().requires_grad_()
            y = ctx.op(xr, wr)
            gx, gw = torch.autograd.grad(y, (xr, wr), _q(gy))
        return gx, gw, None


def _conv(x, w, bias=None, stride=1, padding=0, dilation=1):
    if _QUANT is None:
        return F.conv2d(x, w, bias, stride=stride, padding=padding, dilation=dilation)
    y = _QuantGemmOp.apply(x, w, lambda a, b: F.conv2d(a, b, None, stride=stride, padding=padding, dilation=dilation))
    return y if bias is None else y + bias.view(1, -1, 1, 1)


def _deconv(x, w, stride=1, padding=0):
    if _QUANT is None:
        return F.conv_transpose2d(x, w, stride=stride, padding=padding)
    return _QuantGemmOp.apply(x, w, lambda a, b: F.conv_transpose2d(a, b, stride=stride, padding=padding))


# Pinned decisions (forward_loss(decisions=...)): the discrete choices of a forward pass -- which ReLU inputs are
# positive, which window element a max-pool takes -- as some OTHER evaluation of the same graph made them (the device,
# in the graph parity tests).  A float32 and a float64 forward disagree on the sign of the ~1e-4 of pre-activations that
# lie within rounding of zero, and every such flip moves single gradient entries by O(1); with the decisions pinned the
# two backward passes differentiate the SAME piecewise-linear function and can be compared element by element.
#   "relu:<layer>"  bool (N,C,H,W): the layer's ReLU passes the element       (<layer> = BatchNorm / Convolution name)
#   "pool:<i>"      int64 (N,C,Ho,Wo): window position r*k+s of the maximum of the i-th max-pool in forward order
_DECISIONS = None
_POOL_CALLS = [0]


def _relu(y, name):
    if _DECISIONS is not None and name is not None and ("relu:" + name) in _DECISIONS:
        return y * _DECISIONS["relu:" + name].to(y.dtype)
    return F.relu(y)


def _max_pool(x, k, s, p=0, ceil_mode=False):
    i = _POOL_CALLS[0]
    _POOL_CALLS[0] += 1
    key = "pool:%d" % i
    if _DECISIONS is None or key not in _DECISIONS:
        return F.max_pool2d(x, k, s, p, ceil_mode=ceil_mode)
    pos = _DECISIONS[key]
    N, C, Ho, Wo = pos.shape
    H, W = x.shape[2], x.shape[3]
    oh = torch.arange(Ho).view(1, 1, Ho, 1)
    ow = torch.arange(Wo).view(1, 1, 1, Wo)
    ih = oh * s - p + torch.div(pos, k, rounding_mode="floor")
    iw = ow * s - p + pos % k
    assert int(ih.min()) >= 0 and int(ih.max()) < H and int(iw.min()) >= 0 and int(iw.max()) < W
    return x.flatten(2).gather(2, (ih * W + iw).flatten(2)).view(N, C, Ho, Wo)


def bn(x, gamma, beta, relu=False, name=None):
    mean = x.mean(dim=(0, 2, 3), keepdim=True)
    var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    y = (x - mean) / torch.sqrt(var + EPS)
    if gamma is not None:
        y = y * gamma.view(1, -1, 1, 1)
    y = y + beta.view(1, -1, 1, 1)
    return _relu(y, name) if relu else y


class Params:
    """name -> torch tensor (NCHW-style shapes) with requires_grad; built from the device graph's
    parameters (device layout [Cout,R,S,Cin_phys] -> [Cout,Cin,R,S])."""

    def __init__(self, values, dtype):
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in values.items()}

    def __getitem__(self, k):
        return self.p[k]

    def get(self, k):
        return self.p.get(k)


def residual_unit(P, data, name, stride, dim_match):
    act1 = bn(data, P[name + "_bn1_gamma"], P[name + "_bn1_beta"], relu=True, name=name + "_bn1")
    conv1 = _conv(act1, P[name + "_conv1_weight"])
    act2 = bn(conv1, P[name + "_bn2_gamma"], P[name + "_bn2_beta"], relu=True, name=name + "_bn2")
    conv2 = _conv(act2, P[name + "_conv2_weight"], stride=stride, padding=1)
    act3 = bn(conv2, P[name + "_bn3_gamma"], P[name + "_bn3_beta"], relu=True, name=name + "_bn3")
    conv3 = _conv(act3, P[name + "_conv3_weight"])
    shortcut = data if dim_match else _conv(act1, P[name + "_sc_weight"], stride=stride)
    return conv3 + shortcut


def resnet50(P, data):
    x = bn(data, None, P["bn_data_beta"])
    x = _conv(x, P["conv0_weight"], stride=2, padding=3)
    x = bn(x, P["bn0_gamma"], P["bn0_beta"], relu=True, name="bn0")
    body = _max_pool(x, 3, 2, 1)
    internals, plus = {}, 0
    for i, n in enumerate([3, 4, 6, 3]):
        for j in range(n):
            body = residual_unit(P, body, "stage%d_unit%d" % (i + 1, j + 1), (1 if i == 0 else 2) if j == 0 else 1,
                                 j != 0)
            internals["_plus%d" % plus] = body
            plus += 1
    return internals


def conv_bn(P, x, name, pad):
    return bn(_conv(x, P[name + "_weight"], padding=pad), None, P[name + "_bn_beta"])


def vgg16_reduced(P, data):
    """symbol/vgg16_reduced.py:3-75"""
    def c(x, name, pad=1, dil=1):
        return _relu(_conv(x, P[name + "_weight"], P[name + "_bias"], padding=pad, dilation=dil), name)
    inter = {}
    x = c(c(data, "conv1_1"), "conv1_2")
    x = _max_pool(x, 2, 2)
    x = c(c(x, "conv2_1"), "conv2_2")
    x = _max_pool(x, 2, 2)
    x = c(c(c(x, "conv3_1"), "conv3_2"), "conv3_3")
    x = _max_pool(x, 2, 2, ceil_mode=True)          # pooling_convention="full"
    x = c(c(c(x, "conv4_1"), "conv4_2"), "conv4_3")
    inter["relu4_3"] = x
    x = _max_pool(x, 2, 2)
    x = c(c(c(x, "conv5_1"), "conv5_2"), "conv5_3")
    x = _max_pool(x, 3, 1, 1)
    x = c(x, "fc6", pad=6, dil=6)
    x = c(x, "fc7", pad=0)
    inter["relu7"] = x
    return inter


def inceptionv3(P, data):
    """symbol/inceptionv3.py:10-160: Conv = conv(no bias) -> BN(fix_gamma, MXNet default eps 1e-3) -> ReLU"""
    def C(x, name, suffix='', stride=1, pad=(0, 0)):
        n = '%s%s' % (name, suffix)
        y = _conv(x, P[n + "_conv2d_weight"], stride=stride, padding=pad)
        mean = y.mean(dim=(0, 2, 3), keepdim=True)
        var = y.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        return _relu((y - mean) / torch.sqrt(var + 1e-3) + P[n + "_batchnorm_beta"].view(1, -1, 1, 1), n + "_batchnorm")

    def pool(x, kind, k, s, p):
        return _max_pool(x, k, s, p) if kind == "max" else F.avg_pool2d(x, k, s, p, count_include_pad=True)

    def A(x, kind, name):
        t1 = C(x, name + '_conv')
        t5 = C(C(x, name + '_tower', '_conv'), name + '_tower', '_conv_1', pad=(2, 2))
        t3 = C(C(C(x, name + '_tower_1', '_conv'), name + '_tower_1', '_conv_1', pad=(1, 1)), name + '_tower_1', '_conv_2',
               pad=(1, 1))
        return torch.cat([t1, t5, t3, C(pool(x, kind, 3, 1, 1), name + '_tower_2', '_conv')], dim=1)

    def Bk(x, name):
        t3 = C(x, name + '_conv', stride=2)
        td = C(C(C(x, name + '_tower', '_conv'), name + '_tower', '_conv_1', pad=(1, 1)), name + '_tower', '_conv_2', stride=2)
        return torch.cat([t3, td, pool(x, "max", 3, 2, 0)], dim=1)

    def Ck(x, kind, name):
        t1 = C(x, name + '_conv')
        td = C(C(C(x, name + '_tower', '_conv'), name + '_tower', '_conv_1', pad=(0, 3)), name + '_tower', '_conv_2', pad=(3, 0))
        tq = C(x, name + '_tower_1', '_conv')
        for sfx, pd in (('_conv_1', (3, 0)), ('_conv_2', (0, 3)), ('_conv_3', (3, 0)), ('_conv_4', (0, 3))):
            tq = C(tq, name + '_tower_1', sfx, pad=pd)
        return torch.cat([t1, td, tq, C(pool(x, kind, 3, 1, 1), name + '_tower_2', '_conv')], dim=1)

    def Dk(x, kind, name):
        t3 = C(C(x, name + '_tower', '_conv'), name + '_tower', '_conv_1', stride=2)
        td = C(x, name + '_tower_1', '_conv')
        td = C(C(td, name + '_tower_1', '_conv_1', pad=(0, 3)), name + '_tower_1', '_conv_2', pad=(3, 0))
        td = C(td, name + '_tower_1', '_conv_3', stride=2)
        return torch.cat([t3, td, pool(x, kind, 3, 2, 0)], dim=1)

    def Ek(x, kind, name):
        t1 = C(x, name + '_conv')
        td = C(x, name + '_tower', '_conv')
        ta, tb = C(td, name + '_tower', '_mixed_conv', pad=(0, 1)), C(td, name + '_tower', '_mixed_conv_1', pad=(1, 0))
        t3 = C(C(x, name + '_tower_1', '_conv'), name + '_tower_1', '_conv_1', pad=(1, 1))
        t3a, t3b = C(t3, name + '_tower_1', '_mixed_conv', pad=(0, 1)), C(t3, name + '_tower_1', '_mixed_conv_1', pad=(1, 0))
        return torch.cat([t1, ta, tb, t3a, t3b, C(pool(x, kind, 3, 1, 1), name + '_tower_2', '_conv')], dim=1)

    inter = {}
    x = C(C(C(data, "conv", stride=2), "conv_1"), "conv_2", pad=(1, 1))
    x = _max_pool(x, 3, 2)
    x = C(C(x, "conv_3"), "conv_4")
    x = _max_pool(x, 3, 2)
    x = A(x, "avg", "mixed"); x = A(x, "avg", "mixed_1"); x = A(x, "avg", "mixed_2")
    x = Bk(x, "mixed_3")
    for nm in ("mixed_4", "mixed_5", "mixed_6", "mixed_7"):
        x = Ck(x, "avg", nm)
    inter["ch_concat_mixed_7_chconcat"] = x
    x = Dk(x, "max", "mixed_8")
    x = Ek(x, "avg", "mixed_9")
    x = Ek(x, "max", "mixed_10")
    inter["ch_concat_mixed_10_chconcat"] = x
    return inter


def forward_loss(values, data, label_det, label_seg, sizes=None, ratios=None, num_classes=8, dtype=torch.float64,
                 nms_thresh=0.5, force_suppress=False, nms_topk=400, targets=None, config=None, with_seg=True,
                 conv_quant=None, decisions=None, with_det=True):
    """Runs the multi-task (or, with_seg=False, the detection+depth; with_det=False, the segmentation-only,
    multitask_symbol_builder.py:211-323) training graph on the CPU.
    `config` is the preset of multitask_symbol_factory.get_config (un-sliced); without it the resnet-50
    preset wiring is assumed and sizes/ratios are the already sliced lists.  Returns dict with the graph
    outputs, the loss readouts, the scalar objective whose gradient MXNet's loss ops inject, and Params.
    conv_quant="bf16": bf16-operand convolutions (see _QUANT above); call .backward() on the objective INSIDE
    `with quantized("bf16"):` as well, the gradient GEMMs read the same switch.
    decisions: pinned ReLU / max-pool choices (see _DECISIONS above; every backbone)."""
    global _QUANT, _DECISIONS
    prev, _QUANT = _QUANT, conv_quant
    prev_d, _DECISIONS = _DECISIONS, decisions
    _POOL_CALLS[0] = 0
    try:
        return _forward_loss(values, data, label_det, label_seg, sizes, ratios, num_classes, dtype, nms_thresh,
                             force_suppress, nms_topk, targets, config, with_seg, with_det)
    finally:
        _QUANT, _DECISIONS = prev, prev_d


class quantized:
    """context manager: `with quantized("bf16"): ref["objective"].backward()`"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        global _QUANT
        self.prev, _QUANT = _QUANT, self.mode

    def __exit__(self, *a):
        global _QUANT
        _QUANT = self.prev


def _forward_loss(values, data, label_det, label_seg, sizes, ratios, num_classes, dtype, nms_thresh, force_suppress,
                  nms_topk, targets, config, with_seg, with_det=True):
    P = Params(values, dtype)
    x = torch.tensor(data, dtype=dtype)
    B, _, H, W = x.shape
    if config is None:
        config = dict(network="resnet", from_layers=['_plus6', '_plus12', '_plus15', '', '', '', ''],
                      num_filters=[-1, -1, -1, 512, 256, 256, 128], strides=[-1, -1, -1, 2, 2, 2, 2],
                      pads=[-1, -1, -1, 1, 1, 1, 1], sizes=[None] + list(sizes), ratios=[None] + list(ratios), steps=[])
    inter = {"resnet": resnet50, "vgg16_reduced": vgg16_reduced, "inceptionv3": inceptionv3}[config["network"]](P, x)
    fl = config["from_layers"]
    res3, res4 = inter[fl[0]], inter[fl[1]]
    if not with_det:      # segmentation only: conv_feat is the third backbone map (:269), nothing of SSD exists
        out = dict(params=P, objective=0.0)
        return _segmentation_branch(out, P, inter[fl[2]], res3, res4, label_seg, B, H, W, dtype)
    fl, nfs, sts, pds = fl[1:], config["num_filters"][1:], config["strides"][1:], config["pads"][1:]
    sizes, ratios = config["sizes"][1:], config["ratios"][1:]
    steps = list(config.get("steps") or [])[1:]

    # SSD extras + heads (symbol/common.py:117-133, 393-432)
    layers, names = [], []
    for k, (name, nf, st, pd) in enumerate(zip(fl, nfs, sts, pds)):
        if name:
            layers.append(inter[name]); names.append(name)
        else:
            n1, n3 = "multi_feat_%d_conv_1x1_conv" % k, "multi_feat_%d_conv_3x3_conv" % k
            c1 = _relu(_conv(layers[-1], P[n1 + "_weight"], P[n1 + "_bias"]), n1)
            c3 = _relu(_conv(c1, P[n3 + "_weight"], P[n3 + "_bias"], stride=st, padding=pd), n3)
            layers.append(c3); names.append("multi_feat_%d_conv_3x3_relu" % k)
    conv_feat = layers[1]
    locs, clss, anchors = [], [], []
    for k, (layer, nm, sz, rt) in enumerate(zip(layers, names, sizes, ratios)):
        lp = _conv(layer, P[nm + "_loc_pred_conv_weight"], P[nm + "_loc_pred_conv_bias"], padding=1)
        cp = _conv(layer, P[nm + "_cls_pred_conv_weight"], P[nm + "_cls_pred_conv_bias"], padding=1)
        locs.append(lp.permute(0, 2, 3, 1).reshape(B, -1))
        clss.append(cp.permute(0, 2, 3, 1).reshape(B, -1))
        st = (steps[k], steps[k]) if steps else (-1.0, -1.0)
        anchors.append(om.multibox_prior(layer.shape[2], layer.shape[3], sz, rt, steps=st))
    loc_preds = torch.cat(locs, dim=1)
    ncls = num_classes + 1
    cls_preds = torch.cat(clss, dim=1).reshape(B, -1, ncls).permute(0, 2, 1)       # (B, C+1, N)
    anchor_boxes = np.concatenate(anchors, axis=1)

    if targets is None:
        loc_t, loc_m, cls_t = om.multibox_target(anchor_boxes, label_det, cls_preds.detach().float().numpy(),
                                                 overlap_threshold=.5, ignore_label=-1, negative_mining_ratio=3,
                                                 minimum_negative_samples=0, negative_mining_thresh=.5)
    else:   # graph-level numerics checks pin the (discrete) matching to the device's, which is
        loc_t, loc_m, cls_t = targets   # itself checked bit-exactly against the C oracle
    cls_prob = torch.softmax(cls_preds, dim=1)
    ct = torch.tensor(cls_t, dtype=torch.long)
    valid = ct >= 0
    nvalid = max(1, int(valid.sum()))
    logp = torch.log_softmax(cls_preds, dim=1)
    picked = logp.gather(1, ct.clamp(min=0).unsqueeze(1)).squeeze(1)
    obj_cls = -(picked * valid.to(dtype)).sum() / nvalid
    d = torch.tensor(loc_m, dtype=dtype) * (loc_preds - torch.tensor(loc_t, dtype=dtype))
    loc_loss = torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5)
    nloc = max(1, int((loc_loss > 0).sum()))
    obj_loc = loc_loss.sum() / nloc
    det = om.multibox_detection(cls_prob.detach().float().numpy(), loc_preds.detach().float().numpy(),
                                anchor_boxes, nms_threshold=nms_thresh, force_suppress=force_suppress,
                                nms_topk=nms_topk)

    ce = float(-(torch.log(cls_prob.detach().gather(1, ct.clamp(min=0).unsqueeze(1)).squeeze(1) + 1e-8)
                 * valid.to(dtype)).sum() / nvalid)
    out = dict(cls_prob=cls_prob.detach(), loc_loss=loc_loss.detach(), cls_label=cls_t, det=det, CrossEntropy=ce,
               SmoothL1=float(loc_loss.detach().sum()) / nvalid, params=P, anchors=anchor_boxes,
               loc_preds=loc_preds.detach(), cls_preds=cls_preds.detach())
    out["objective"] = obj_cls + obj_loc
    if not with_seg:
        return out
    return _segmentation_branch(out, P, conv_feat, res3, res4, label_seg, B, H, W, dtype)


def _segmentation_branch(out, P, conv_feat, res3, res4, label_seg, B, H, W, dtype):
    # segmentation decoder (multitask_symbol_builder.py:541-589)
    r3 = conv_bn(P, conv_bn(P, res3.detach(), "res3_reduced", 0), "res3_reduced2", 1)
    r4 = conv_bn(P, conv_bn(P, res4.detach(), "res4_reduced", 0), "res4_reduced2", 1)
    r5 = bn(conv_feat, None, P["res5_reduced_bn_beta"])
    p4 = conv_bn(P, F.avg_pool2d(r5, 4, 4), "score2_pool4", 0)
    p2 = conv_bn(P, F.avg_pool2d(r5, 2, 2), "score2_pool2", 0)
    p1 = conv_bn(P, r5, "score2_pool1", 0)
    th, tw = H // 8, W // 8
    theta = P["affine_matrix"].reshape(1, 2, 3).expand(B, 2, 3)
    grid = F.affine_grid(theta, (B, 1, th, tw), align_corners=True)
    samp = [F.grid_sample(t, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
            for t in (p4, p2, p1, r5, r4, r3)]
    if _QUANT is None:
        cat = torch.cat(samp, dim=1)
        s3 = conv_bn(P, cat, "score3_conv", 1)
    else:
        # The device evaluates score3_conv per pyramid level BEFORE the resize (engine.BilinearConcatConv: an exact
        # identity in real arithmetic), so its bf16 operand rounding happens on the un-resized maps.  Same here: the
        # tap-expanded 1x1 products at each level's own resolution, sampled, summed, then the shifted sum over taps.
        Wf = P["score3_conv_weight"]                                   # (19, 3328, 3, 3)
        co, _, kh, kw = Wf.shape
        z, off = 0, 0
        for t in (p4, p2, p1, r5, r4, r3):
            wc = Wf[:, off:off + t.shape[1]].permute(0, 2, 3, 1).reshape(co * kh * kw, t.shape[1], 1, 1)   # row = (co, r, s)
            off += t.shape[1]
            z = z + F.grid_sample(_conv(t, wc), grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        zp = F.pad(z.reshape(B, co, kh, kw, th, tw), (1, 1, 1, 1))
        s3c = sum(zp[:, :, r, q, r:r + th, q:q + tw] for r in range(kh) for q in range(kw))
        s3 = bn(s3c, None, P["score3_conv_bn_beta"])
    s4 = _deconv(s3, P["score4_conv_weight"], stride=2, padding=1)
    seg_prob = torch.softmax(s4, dim=1)
    sl = torch.tensor(label_seg, dtype=torch.long)
    svalid = sl != 255
    slogp = torch.log_softmax(s4, dim=1).gather(1, sl.clamp(max=18).unsqueeze(1)).squeeze(1)
    seg_ce_sum = -(slogp * svalid.to(dtype)).sum()
    obj_seg = seg_ce_sum * (4.0 / float(s4.shape[2] * s4.shape[3]))
    out["objective"] = out["objective"] + obj_seg
    out["seg_out"] = seg_prob.detach()
    out["SegCrossEntropy"] = float(-(torch.log(seg_prob.detach().gather(1, sl.clamp(max=18).unsqueeze(1)).squeeze(1) + 1e-8)
                                     * svalid.to(dtype)).sum() / max(1, int(svalid.sum())))
    return out


def export_params(graph):
    """device graph parameters -> {name: numpy in NCHW-style logical shapes}"""
    out = {}
    for p in graph.param_order:
        v = p.data.detach().cpu().numpy()
        if p.name == "score4_conv_weight":          # [K=in][R][S][C=out] phys 20 -> (in 19, out 19, 4, 4)
            v = v[:19, :, :, :19].transpose(0, 3, 1, 2)
        elif v.ndim == 4:                            # [Cout][R][S][Cin_phys] -> [Cout][Cin][R][S]
            v = v.transpose(0, 3, 1, 2)
            if p.name in ("conv0_weight", "conv1_1_weight", "conv_conv2d_weight"):
                v = v[:, :3]
            if p.name == "score4_conv_weight":
                pass
        elif p.name in ("score3_conv_bn_beta",):
            v = v[:19]
        elif p.name == "bn_data_beta":
            v = v[:3]
        out[p.name] = np.ascontiguousarray(v)
    return out


def import_grad(name, g):
    """oracle gradient (torch, logical shape) -> device layout numpy for comparison (pads dropped)"""
    g = g.detach().cpu().numpy()
    if name == "affine_matrix":
        return g.reshape(-1)
    if name == "score4_conv_weight":
        return g.transpose(0, 2, 3, 1)               # (in,out,4,4) -> [in][R][S][out]
    if g.ndim == 4:
        return g.transpose(0, 2, 3, 1)
    return g
