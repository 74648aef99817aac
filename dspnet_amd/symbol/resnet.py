"""Pre-activation ResNet backbone (counterpart of symbol/resnet.py of the reference).

Same unit wiring, layer names and hyper-parameters: BN(eps=2e-5, fix_gamma=False) -> ReLU ->
conv, bottleneck 1x1 / 3x3(stride) / 1x1, projection shortcut on act1 when the shape changes
(symbol/resnet.py:30-51); stem bn_data(fix_gamma) -> conv0 7x7/2 -> bn0 -> relu0 -> maxpool 3x3/2
(:89-98).  The classifier tail (:109-116) is not built: get_multi_symbol_train only reads the
`_plusN` internals.  BN+ReLU run as one fused kernel; every conv is the fp32-MFMA implicit GEMM."""
import torch

from .. import engine as E
from .. import functional as fn


import os
MATERIALISE_3X3_INPUT = os.environ.get("DSPN_MAT3X3", "1") != "0"    # test / A-B switch (bf16 tensors only)
MATERIALISE_3X3_INPUT_F32 = os.environ.get("DSPN_MAT3X3_F32", "0") != "0"   # A-B switch: the same for float tensors


def residual_unit(g, data, num_filter, stride, dim_match, name, plus_name, bottle_neck=True):
    """symbol/resnet.py:11-68"""
    # The unit's `+ shortcut` (symbol/resnet.py:51,67) is folded into the epilogue of its last convolution,
    # whose output tensor therefore IS `_plusN`; the projection shortcut is built before that conv.
    if not bottle_neck:
        bn1 = g.add(E.BatchNorm(g, data, name + "_bn1", relu=True, defer_apply=True)).out
        shortcut = data if dim_match else g.add(E.Conv(g, bn1, name + "_sc", num_filter, 1, stride, 0)).out
        conv1 = g.add(E.Conv(g, bn1, name + "_conv1", num_filter, 3, stride, 1)).out
        bn2 = g.add(E.BatchNorm(g, conv1, name + "_bn2", relu=True, defer_apply=True)).out
        return g.add(E.Conv(g, bn2, name + "_conv2", num_filter, 3, 1, 1, residual=shortcut, out_name=plus_name)).out
    q = int(num_filter * 0.25)
    # act1/act2/act3 feed convolutions only: their BN-apply + ReLU runs inside those convolutions' loaders
    act1 = g.add(E.BatchNorm(g, data, name + "_bn1", relu=True, defer_apply=True)).out
    # conv1 is created before the projection shortcut so that its (dense, stride-1) data gradient is the last
    # writer of act1's gradient in backward and can gather bn1's backward reductions (engine._plan_bn_backward_fusion)
    conv1 = g.add(E.Conv(g, act1, name + "_conv1", q, 1, 1, 0)).out
    shortcut = data if dim_match else g.add(E.Conv(g, act1, name + "_sc", num_filter, 1, stride, 0)).out
    # bf16 tensors: the 3x3 convolution would re-apply the folded BatchNorm+ReLU to every element once per tap (9x) on a
    # main loop that is 3x shorter than in fp32 (+45 % on these layers, scratch/fuse_cost.py bf16), while act2 is the
    # smallest tensor of the unit -- materialising it costs one 4-byte-per-element pass.  Its backward reductions still
    # come out of conv2's data-gradient epilogue.
    act2 = g.add(E.BatchNorm(g, conv1, name + "_bn2", relu=True,
                             defer_apply=not (MATERIALISE_3X3_INPUT and (fn.ACT_DTYPE == torch.bfloat16 or MATERIALISE_3X3_INPUT_F32)))).out
    conv2 = g.add(E.Conv(g, act2, name + "_conv2", q, 3, stride, 1)).out
    act3 = g.add(E.BatchNorm(g, conv2, name + "_bn3", relu=True, defer_apply=True)).out
    return g.add(E.Conv(g, act3, name + "_conv3", num_filter, 1, 1, 0, residual=shortcut, out_name=plus_name)).out


def resnet(g, data, units, num_stages, filter_list, bottle_neck=True):
    """symbol/resnet.py:70-116 without the classifier tail; returns {internal name: Tensor}"""
    internals = {}
    x = g.add(E.InputNCHW(g, data)).out
    # bn_data's only trainable parameter is beta, whose gradient is sum_pixels(d conv0 / d input): conv0
    # computes that sum directly instead of a 3-channel data gradient
    bn_data = g.add(E.BatchNorm(g, x, "bn_data", fix_gamma=True, beta_grad_from_consumer=True))
    x = g.add(E.Conv(g, bn_data.out, "conv0", filter_list[0], 7, 2, 3, cin_logical=3,
                     input_sum_grad=bn_data.beta)).out
    # bn0 -> relu0 -> pooling0 (symbol/resnet.py:96-98): the BatchNorm-apply + ReLU runs inside the pooling pass (round 4), the
    # normalised 64-channel 256 x 256 tensor is never written (engine.FUSE_BATCHNORM = False materialises it again)
    x = g.add(E.BatchNorm(g, x, "bn0", relu=True, defer_apply=True)).out
    body = g.add(E.MaxPool(g, x, "pooling0", 3, 2, 1)).out
    plus = 0
    for i in range(num_stages):
        s = 1 if i == 0 else 2
        body = residual_unit(g, body, filter_list[i + 1], s, False, "stage%d_unit%d" % (i + 1, 1),
                             "_plus%d" % plus, bottle_neck)
        internals["_plus%d_output" % plus] = body
        plus += 1
        for j in range(units[i] - 1):
            body = residual_unit(g, body, filter_list[i + 1], 1, True, "stage%d_unit%d" % (i + 1, j + 2),
                                 "_plus%d" % plus, bottle_neck)
            internals["_plus%d_output" % plus] = body
            plus += 1
    internals["_output"] = body
    return internals


def get_symbol(g, data, num_layers=50, **kwargs):
    """symbol/resnet.py:118-169 (ImageNet-style configurations)"""
    if num_layers >= 50:
        filter_list, bottle_neck = [64, 256, 512, 1024, 2048], True
    else:
        filter_list, bottle_neck = [64, 64, 128, 256, 512], False
    table = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3],
             200: [3, 24, 36, 3], 269: [3, 30, 48, 8]}
    if num_layers not in table:
        raise ValueError("no experiments done on num_layers {}, you can do it yourself".format(num_layers))
    return resnet(g, data, table[num_layers], 4, filter_list, bottle_neck)
