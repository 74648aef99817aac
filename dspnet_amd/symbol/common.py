"""SSD feature / head builders (counterpart of symbol/common.py of the reference)."""
import numpy as np

from .. import engine as E
from .. import functional as fn
from .. import operator as op


def conv_act_layer(g, from_layer, name, num_filter, kernel=1, pad=0, stride=1):
    """symbol/common.py:4-38: Convolution (with bias) + ReLU, fused into the conv epilogue"""
    return g.add(E.Conv(g, from_layer, "{}_conv".format(name), num_filter, kernel, stride, pad,
                        no_bias=False, relu=True, init="maxdim")).out


def multi_layer_feature(g, internals, from_layers, num_filters, strides, pads, min_filter=128):
    """symbol/common.py:79-134"""
    assert len(from_layers) > 0
    assert isinstance(from_layers[0], str) and len(from_layers[0].strip()) > 0
    assert len(from_layers) == len(num_filters) == len(strides) == len(pads)
    layers = []
    for k, (from_layer, num_filter, s, p) in enumerate(zip(from_layers, num_filters, strides, pads)):
        if from_layer.strip():
            t = internals[from_layer.strip() + "_output"]
            t.ssd_name = from_layer.strip()          # the symbol name the reference derives head names from
            layers.append(t)
        else:
            assert len(layers) > 0 and num_filter > 0
            num_1x1 = max(min_filter, num_filter // 2)
            c1 = conv_act_layer(g, layers[-1], "multi_feat_%d_conv_1x1" % k, num_1x1, 1, 0, 1)
            c3 = conv_act_layer(g, c1, "multi_feat_%d_conv_3x3" % k, num_filter, 3, p, s)
            c3.ssd_name = "multi_feat_%d_conv_3x3_relu" % k
            layers.append(c3)
    return layers


class HeadPack(E.Node):
    """transpose(0,2,3,1) + Flatten + Concat(dim=1) of the per-map predictions
    (symbol/common.py:396-412,424-426).  The convs already produce NHWC, so this only strips the
    channel padding and packs each map at its offset of the (B, sum_k H_k*W_k*Ck) row."""

    def __init__(self, g, maps, widths, name):
        self.maps, self.widths = maps, widths
        B = maps[0].shape[0]
        self.sizes = [t.shape[1] * t.shape[2] * c for t, c in zip(maps, widths)]
        self.offsets = np.cumsum([0] + self.sizes).tolist()
        self.total = self.offsets[-1]
        import torch
        self.out = g.tensor((B, self.total), name, dtype=torch.float32)     # loss / multibox-operator input: always float
        self._tables = {}

    def _batched(self, key, entries):
        """float maps on the GPU: the copies of one pass as ONE launch (round 4; a table per distinct set of buffers)"""
        import torch
        if self.out.data.device.type != "cuda" or any(e[0].dtype != torch.float32 or e[1].dtype != torch.float32 for e in entries):
            return False
        key = (key,) + tuple((e[0].data_ptr(), e[1].data_ptr(), bool(e[-1])) for e in entries)
        tab = self._tables.get(key)
        if tab is None:
            tab = self._tables[key] = fn.copy_block_table(entries, self.out.data.device)
        fn.copy_block_batch(*tab)
        return True

    def forward(self):
        B = self.out.shape[0]
        entries = [(t.data, self.out.data, B, t.shape[1] * t.shape[2], c, t.shape[1] * t.shape[2] * t.shape[3], t.shape[3], 0,
                    self.total, c, off, False) for t, c, off in zip(self.maps, self.widths, self.offsets)]
        if not self._batched("f", entries):
            for e in entries:
                fn.copy_block(*e[:-1])

    def backward(self):
        if not self.out._gw:
            return
        B = self.out.shape[0]
        entries = []
        for t, c, off in zip(self.maps, self.widths, self.offsets):
            hw = t.shape[1] * t.shape[2]
            dx, acc = t.grad_target()          # pad channels of dx stay zero (allocated zeroed)
            entries.append((self.out.grad, dx, B, hw, c, self.total, c, off, hw * t.shape[3], t.shape[3], 0, acc))
        if not self._batched("b", entries):
            for e in entries:
                fn.copy_block(*e[:-1], accumulate=e[-1])


def multitask_layer(g, from_layers, num_classes, sizes, ratios, normalization=-1, clip=False, steps=()):
    """symbol/common.py:286-433 for the configurations DSPNet uses (no L2Normalization, no
    intermediate conv).  Returns (loc_preds (B, N*5), cls_flat (B, N*(C+1)), anchors (1, N, 4))."""
    assert len(from_layers) > 0, "from_layers must not be empty list"
    assert num_classes > 0, "num_classes {} must be larger than 0".format(num_classes)
    assert len(ratios) == len(from_layers), "ratios and from_layers must have same length"
    assert len(sizes) == len(from_layers), "sizes and from_layers must have same length"
    if not isinstance(normalization, (list, tuple)):
        normalization = [normalization] * len(from_layers)
    assert all(n <= 0 for n in normalization), "L2Normalization heads are not part of the DSPNet presets that build"
    num_classes += 1
    loc_maps, cls_maps, loc_w, cls_w, anchors = [], [], [], [], []
    for k, from_layer in enumerate(from_layers):
        from_name = from_layer.ssd_name              # symbol/common.py:367 `from_layer.name`
        size, ratio = sizes[k], ratios[k]
        num_anchors = len(size) - 1 + len(ratio)
        # 20..54 output channels starve the 32-wide MFMA column dimension: on the large maps the 3x3 head
        # convolutions run tap-expanded (a 1x1 convolution to Cout*9 channels + a shifted sum, engine.Conv)
        big = from_layer.shape[0] * from_layer.shape[1] * from_layer.shape[2] >= 4096
        loc = g.add(E.Conv(g, from_layer, "{}_loc_pred_conv".format(from_name), num_anchors * 5, 3, 1, 1,
                           no_bias=False, init="maxdim", tap_expand=big)).out
        cls = g.add(E.Conv(g, from_layer, "{}_cls_pred_conv".format(from_name), num_anchors * num_classes, 3,
                           1, 1, no_bias=False, init="maxdim", tap_expand=big)).out
        loc_maps.append(loc); loc_w.append(num_anchors * 5)
        cls_maps.append(cls); cls_w.append(num_anchors * num_classes)
        step = (steps[k], steps[k]) if steps else (-1.0, -1.0)
        # anchors depend on the map shape only: generated once at build time
        anchors.append(op.MultiBoxPrior((from_layer.shape[1], from_layer.shape[2]), sizes=size, ratios=ratio,
                                        clip=clip, steps=step))
    loc_preds = g.add(HeadPack(g, loc_maps, loc_w, "multibox_loc_pred")).out
    cls_flat = g.add(HeadPack(g, cls_maps, cls_w, "multibox_cls_flat")).out
    import torch
    anchor_boxes = torch.cat(anchors, dim=1).contiguous()
    return loc_preds, cls_flat, anchor_boxes
