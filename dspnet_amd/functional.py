"""Thin functional layer over the C ABI of include/dspn_nn.h.

torch tensors are device buffers only; every function launches HIP kernels on
torch's current stream.  Activations are NHWC float32 with a physical channel
count that is a multiple of 4; conv weights are [Cout, R, S, Cin]."""
import ctypes
import os

import torch

from . import _lib
from ._lib import check

_c = ctypes
_vp = _c.c_void_p
_i = _c.c_int
_ll = _c.c_longlong
_f = _c.c_float
_sz = _c.c_size_t

_lib.register({
    "dspn_conv2d_split_workspace_bytes": (_sz, [_ll, _i]),
    "dspn_conv2d_forward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                     _ll, _i, _i, _i, _vp, _sz, _vp]),
    "dspn_conv2d_forward_bn_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                        _i, _ll, _i, _i, _i, _vp, _sz, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "dspn_absmax_f32": (_i, [_vp, _ll, _i, _vp, _vp, _i, _vp, _vp]),
    "dspn_absmax_batch_f32": (_i, [_vp, _i, _ll, _vp]),
    "dspn_conv2d_weight_planes_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "dspn_conv2d_weight_planes_tiles": (_ll, [_i, _i, _i, _i, _i]),
    "dspn_conv2d_weight_planes_batch_f32": (_i, [_vp, _i, _ll, _vp]),
    "dspn_conv2d_stats_layout": (_i, [_ll, _i, _c.POINTER(_c.c_int)]),
    "dspn_bn_stats_from_tiles_f32": (_i, [_vp, _i, _i, _ll, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dspn_bn_tiles_workspace_bytes": (_sz, [_i, _i]),
    "dspn_conv2d_wgrad_bn_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                      _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "dspn_conv2d_wgrad_splits": (_i, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "dspn_conv2d_wgrad_slabs_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                         _i, _i, _i, _vp, _vp, _vp]),
    "dspn_conv2d_slab_reduce_batch_f32": (_i, [_vp, _i, _ll, _vp]),
    "dspn_conv2d_weight_transpose_batch_f32": (_i, [_vp, _i, _ll, _vp]),
    "dspn_conv2d_weight_transpose_tiles": (_ll, [_i, _i, _i, _i]),
    "dspn_conv2d_weight_transpose_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "dspn_conv2d_dgrad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp,
                                   _sz, _vp]),
    "dspn_conv2d_dgrad_bn_tiles": (_i, [_i, _i, _i, _i, _i]),
    "dspn_conv2d_dgrad_bn_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                      _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "dspn_bn_backward_from_sums_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _ll, _i, _i, _i,
                                            _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "dspn_absmin_rows_batch_f32": (_i, [_vp, _i, _ll, _vp]),
    "dspn_tile_minmax_f32": (_i, [_vp, _ll, _i, _i, _vp, _vp]),
    "dspn_conv2d_input_sum_grad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "dspn_conv2d_input_sum_grad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                            _vp, _sz, _vp]),
    "dspn_conv2d_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "dspn_conv2d_wgrad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                   _vp, _sz, _vp]),
    "dspn_bn_workspace_bytes": (_sz, [_ll, _i]),
    "dspn_bn_stats_f32": (_i, [_vp, _ll, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dspn_bn_apply_f32": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp]),
    "dspn_bn_apply_planes_f32": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp, _vp]),
    "dspn_absmax_affine_bound_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "dspn_bn_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp, _vp, _sz,
                                  _vp]),
    "dspn_add_f32": (_i, [_vp, _vp, _vp, _ll, _vp]),
    "dspn_relu_backward_f32": (_i, [_vp, _vp, _vp, _ll, _i, _vp]),
    "dspn_fill_f32": (_i, [_vp, _f, _ll, _vp]),
    "dspn_relu_backward_colsum_f32": (_i, [_vp, _vp, _vp, _ll, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "dspn_colsum_workspace_bytes": (_sz, [_ll, _i]),
    "dspn_colsum_f32": (_i, [_vp, _ll, _i, _i, _vp, _vp, _sz, _vp]),
    "dspn_nchw_to_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dspn_nhwc_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dspn_copy_block_f32": (_i, [_vp, _vp, _i, _ll, _i, _ll, _i, _i, _ll, _i, _i, _i, _vp]),
    "dspn_copy_block_batch_f32": (_i, [_vp, _i, _ll, _vp]),
    "dspn_transpose_bnc_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "dspn_avgpool2d_forward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_avgpool2d_backward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_bilinear_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dspn_bilinear_backward_ws_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "dspn_seg_counts_f32": (_i, [_vp, _vp, _ll, _i, _i, _vp, _vp]),
    "dspn_seg_upsample_argmax_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_tap_sum_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_tap_spread_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_maxpool_forward_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_maxpool_forward_bn_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "dspn_maxpool_backward_argmax_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_bn_backward_maxpool_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _i, _vp, _vp, _sz, _vp]),
    "dspn_maxpool_backward_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_avgpool_forward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_avgpool_backward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_bilinear_forward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_bilinear_forward_acc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_bilinear_backward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_affine_sampler_forward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dspn_affine_sampler_backward_data_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "dspn_affine_sampler_theta_workspace_bytes": (_sz, [_i, _i, _i]),
    "dspn_affine_sampler_backward_theta_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i,
                                                    _vp, _sz, _vp]),
    "dspn_affine_sampler_backward_data_theta_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp,
                                                         _vp, _sz, _vp]),
    "dspn_affine_sampler_theta_rows": (_ll, [_i, _i, _i, _i]),
    "dspn_affine_sampler_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "dspn_affine_sampler_theta_reduce_workspace_bytes": (_sz, [_ll]),
    "dspn_affine_sampler_theta_reduce": (_i, [_vp, _ll, _vp, _i, _vp, _sz, _vp]),
    "dspn_softmax_output_f32": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _f, _f, _vp, _vp]),
    "dspn_count_f32": (_i, [_vp, _ll, _i, _f, _vp, _vp]),
    "dspn_smooth_l1_forward_f32": (_i, [_vp, _vp, _vp, _vp, _ll, _vp]),
    "dspn_smooth_l1_backward_f32": (_i, [_vp, _vp, _vp, _vp, _ll, _f, _vp, _vp]),
    "dspn_cross_entropy_sum_f32": (_i, [_vp, _vp, _ll, _i, _i, _f, _f, _vp, _vp]),
    "dspn_sum_f32": (_i, [_vp, _ll, _vp, _vp]),
    "dspn_sgd_momentum_f32": (_i, [_vp, _vp, _vp, _ll, _f, _f, _f, _f, _vp]),
})


# the bfloat16-tensor twins (include/dspn_nn.h): same argument lists as the *_f32 entries
for _name in ("dspn_conv2d_forward_bn", "dspn_conv2d_dgrad_bn", "dspn_conv2d_wgrad_bn", "dspn_conv2d_wgrad_slabs",
              "dspn_conv2d_input_sum_grad", "dspn_bn_stats", "dspn_bn_apply", "dspn_bn_backward",
              "dspn_bn_backward_from_sums", "dspn_add", "dspn_relu_backward", "dspn_relu_backward_colsum", "dspn_colsum",
              "dspn_nchw_to_nhwc", "dspn_copy_block", "dspn_tap_sum", "dspn_tap_spread", "dspn_maxpool_forward", "dspn_maxpool_forward_bn",
              "dspn_maxpool_backward_argmax", "dspn_maxpool_backward", "dspn_avgpool_forward", "dspn_avgpool_backward",
              "dspn_avgpool2d_forward", "dspn_avgpool2d_backward", "dspn_softmax_output", "dspn_affine_sampler_forward",
              "dspn_affine_sampler_backward_data", "dspn_affine_sampler_backward_theta",
              "dspn_affine_sampler_backward_data_theta"):
    _lib.register({_name + "_bf16": _lib.SIGNATURES[_name + "_f32"]})
_lib.register({
    "dspn_conv2d_weight_prepare_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dspn_conv2d_weight_prepare_batch_bf16": (_i, [_vp, _i, _ll, _vp]),
    "dspn_copy_block_bf16_f32": _lib.SIGNATURES["dspn_copy_block_f32"],
    "dspn_copy_block_f32_bf16": _lib.SIGNATURES["dspn_copy_block_f32"],
})


def L():
    return _lib.lib()


def _f(name, t):
    """the entry point of `name` for the storage type of activation tensor t (float32 -> *_f32, bfloat16 -> *_bf16)"""
    if t.dtype == torch.bfloat16:
        return getattr(L(), name + "_bf16")
    assert t.dtype == torch.float32, t.dtype
    return getattr(L(), name + "_f32")


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


_ws = {}


_WS_LANE = 0      # 0: the graph's main stream, 1: its side stream (engine.Graph sets it around the nodes it runs there)


class workspace_lane:
    """`with workspace_lane(1):` -- scratch buffers of the calls inside come from another set: two streams of one graph (the
    detection branch beside the segmentation decoder, round 4) must not share scratch.  (Not keyed by the stream itself:
    every solver and every graph makes its own streams, and the buffers of dead ones would pile up.)"""

    def __init__(self, lane):
        self.lane = lane

    def __enter__(self):
        global _WS_LANE
        self.prev, _WS_LANE = _WS_LANE, self.lane

    def __exit__(self, *a):
        global _WS_LANE
        _WS_LANE = self.prev


def workspace(nbytes, device, tag="nn"):
    key = (tag, device, _WS_LANE)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def _hw(v):
    """int or (h, w) pair -> (h, w)"""
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def conv_out_size(h, k, stride, pad, dil=1):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def pad4(c):
    return (c + 3) // 4 * 4


# Storage type of ACTIVATION tensors (and of the convolution operands): torch.float32, or torch.bfloat16 for the
# `*_bf16` kernels.  Parameters, statistics, losses and optimizer state are always float32.  A graph is built under one
# setting (set_activation_dtype before get_multi_symbol_train); every wrapper below picks the kernel from the dtype of
# the tensors it is handed.
ACT_DTYPE = torch.float32


def set_activation_dtype(dtype):
    """"fp32" / torch.float32 or "bf16" / torch.bfloat16 for the activation tensors of graphs built from now on"""
    global ACT_DTYPE
    ACT_DTYPE = {"fp32": torch.float32, "f32": torch.float32, "bf16": torch.bfloat16}.get(dtype, dtype)
    assert ACT_DTYPE in (torch.float32, torch.bfloat16)


def padc(c, dtype=None):
    """physical channel count of an activation tensor: a 16-byte chunk of channels (4 floats / 8 bf16)"""
    return (c + 7) // 8 * 8 if (dtype or ACT_DTYPE) == torch.bfloat16 else pad4(c)


def empty(*shape, device=None, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=device or torch.device("cuda", torch.cuda.current_device()))


def zeros(*shape, device=None, dtype=torch.float32):
    return torch.zeros(shape, dtype=dtype, device=device or torch.device("cuda", torch.cuda.current_device()))


def act_zeros(*shape, device=None):
    return zeros(*shape, device=device, dtype=ACT_DTYPE)


def act_empty(*shape, device=None):
    return empty(*shape, device=device, dtype=ACT_DTYPE)


# ------------------------------------------------------------------ convolution
_MATH_CODES = {"fp32": 0, "f32": 0, "bf16": 1, "bf16x3": 2, "f16x2": 3}      # include/dspn_nn.h DSPN_MATH_*
# this module's default for the `math` argument it passes on every convolution call (float tensors); DSPN_CONV_MATH
# overrides it for a whole process (the test suite is run once per mode)
DEFAULT_CONV_MATH = os.environ.get("DSPN_CONV_MATH", "f16x2")
if DEFAULT_CONV_MATH not in _MATH_CODES:
    raise ValueError("DSPN_CONV_MATH=%r: expected one of %s" % (DEFAULT_CONV_MATH, ", ".join(sorted(_MATH_CODES))))
_MATH = _MATH_CODES[DEFAULT_CONV_MATH]


def set_conv_math(mode):
    """math of the float-tensor convolution calls made through this module from now on (include/dspn_nn.h DSPN_MATH_*):
      "f16x2"   (default since round 3) fp32 results on the fp16 MFMA: two fp16 pieces per operand after a per-tensor
                power-of-two scale, three exact products per multiply -- see below
      "bf16x3"  fp32 results on the bf16 MFMA: every float operand cut into three bf16 pieces on its way into LDS, six
                exact partial products per multiply, fp32 accumulate (the default of round 2; needs no operand magnitudes)
      "fp32"    fp32 MFMA (v_mfma_f32_32x32x2_f32)
      "f16x2"   fp32 results on the fp16 MFMA with half the matrix work of "bf16x3": two fp16 pieces per operand after a
                per-tensor power-of-two scale, three exact products per multiply (include/dspn_nn.h DSPN_MATH_F32_F16X2)
      "bf16"    operands ROUNDED to bf16 (2^-9 relative), fp32 accumulate: BASELINE.json configs[3]
    Host-side default only: the C ABI takes the mode per call, the library itself has no state."""
    global _MATH
    _MATH = _MATH_CODES[mode]


def get_conv_math():
    return ("fp32", "bf16", "bf16x3", "f16x2")[_MATH]


def conv_stats_layout(out_pixels, cout):
    """(tiles, rows per tile) of the per-tile BatchNorm statistics a convolution can emit; (0, 0) if it cannot"""
    tr = _c.c_int(0)
    n = L().dspn_conv2d_stats_layout(int(out_pixels), int(cout), _c.byref(tr))
    return (n, tr.value) if n > 0 else (0, 0)


def bn_stats_from_tiles(tile_stats, tiles, tile_rows, rows, C, eps, gamma, beta, mean, rstd, scale, shift,
                        tile_minmax=None, relu=False, out_absmax=None, out_absmin=None, out_chan_minmax=None):
    """tile_minmax + out_absmax ("f16x2" math): also max the magnitude of (relu)(x * scale + shift) into the 64-slot block
    out_absmax, from the per-tile extremes the producing convolution wrote (conv2d_forward's out_minmax)"""
    ws = workspace(L().dspn_bn_tiles_workspace_bytes(tiles, C), tile_stats.device, "bn")
    assert (tile_minmax is None) == (out_absmax is None)
    assert tile_minmax is None or (tile_minmax.numel() == tile_stats.numel() and out_absmax.numel() == ABSMAX_SLOTS)
    check(L().dspn_bn_stats_from_tiles_f32(ptr(tile_stats), tiles, tile_rows, rows, C, eps, ptr(gamma), ptr(beta), ptr(mean),
                                           ptr(rstd), ptr(scale), ptr(shift), ptr(tile_minmax), int(bool(relu)),
                                           ptr(out_absmax), ptr(out_absmin), ptr(out_chan_minmax), ptr(ws), ws.numel(), stream()),
          "bn_stats_from_tiles")


ABSMAX_SLOTS = 64      # include/dspn_nn.h DSPN_ABSMAX_SLOTS: a magnitude is 64 partial maxima


def absmax(x, in_affine=None, out=None):
    """largest magnitude of the float32 tensor x (last dimension = channels, a multiple of 4) -- of (relu)(x * scale + shift)
    when in_affine = (scale, shift, relu) is given -- as 64 partial maxima in device memory: the `*_absmax` operand of the
    "f16x2" math (`.max()` of the result is the magnitude).  out: an existing 64-element float32 tensor to take the maxima
    INTO (it is not zeroed here)"""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 4 == 0
    if out is None:
        out = zeros(ABSMAX_SLOTS, device=x.device)
    assert out.numel() == ABSMAX_SLOTS and out.dtype == torch.float32
    sc, sh, relu = in_affine if in_affine is not None else (None, None, False)
    C = x.shape[-1]
    check(L().dspn_absmax_f32(ptr(x), x.numel() // C, C, ptr(sc), ptr(sh), int(bool(relu)), ptr(out), stream()), "absmax")
    return out


def absmax_table(pairs, device):
    """pairs: [(float32 tensor, 64-element float32 output)] -> (device table, rows, total chunks) for absmax_batch"""
    import numpy as np
    rows = np.zeros(len(pairs), dtype=[("x", "<u8"), ("out", "<u8"), ("n4", "<i8"), ("begin", "<i8")])
    total = 0
    for i, (t, o) in enumerate(pairs):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0 and o.numel() == ABSMAX_SLOTS
        rows[i] = (t.data_ptr(), o.data_ptr(), t.numel() // 4, total)
        total += (t.numel() // 4 + 1023) // 1024
    assert rows.dtype.itemsize == 32
    return torch.from_numpy(rows.view(np.uint8).copy()).to(device), len(pairs), total


def absmin_rows_table(pairs, device):
    """pairs: [(float32 weight [rows, ...], 1-element float32 output)] -> (device table, rows, total rows) for absmin_rows_batch"""
    import numpy as np
    rows = np.zeros(len(pairs), dtype=[("w", "<u8"), ("out", "<u8"), ("rows", "<i4"), ("row_len", "<i4"), ("begin", "<i8")])
    begin = 0
    for i, (w, out) in enumerate(pairs):
        assert w.dtype == torch.float32 and w.is_contiguous() and out.numel() == 1
        rows[i] = (w.data_ptr(), out.data_ptr(), w.shape[0], w.numel() // w.shape[0], begin)
        begin += w.shape[0]
    table = torch.from_numpy(rows.view(np.uint8).copy()).to(device)
    return table, len(pairs), begin


def absmin_rows_batch(table, n, total):
    """the smallest non-zero per-output-channel magnitude of every weight of the table (outputs preset to +inf by the caller)"""
    check(L().dspn_absmin_rows_batch_f32(ptr(table), n, total, stream()), "absmin_rows_batch")


def tile_minmax(x, tile_rows, out):
    """out (tiles, 2, C): smallest / largest value of each channel over each tile of `tile_rows` rows of x (..., C)"""
    C = x.shape[-1]
    check(L().dspn_tile_minmax_f32(ptr(x), _rows(x), C, int(tile_rows), ptr(out), stream()), "tile_minmax")
    return out


def absmax_batch(table, n, total):
    check(L().dspn_absmax_batch_f32(ptr(table), n, total, stream()), "absmax_batch")


def _math_code(math):
    return _MATH if math is None else (_MATH_CODES[math] if isinstance(math, str) else int(math))


def needs_planes(t_dtype, k_channels, math=None):
    """True when a forward / data-gradient call contracting over k_channels per tap reads its weight operand as piece
    planes (include/dspn_nn.h: the split math modes, float tensors, whole 32-channel blocks)"""
    return t_dtype == torch.float32 and _math_code(math) in (2, 3) and k_channels % 32 == 0


def plane_pieces(math=None):
    """pieces per element of a piece-plane operand: 3 bfloat16 ("bf16x3") or 2 float16 of the scaled value ("f16x2")"""
    return 2 if _math_code(math) == 3 else 3


def weight_planes(w, transposed=False, cols=None, out=None, math=None, w_absmax=None):
    """piece planes (rows, taps, cols / 32, pieces, 32) of the float32 master w (Cout, R, S, Cin): of w itself (rows = Cout,
    cols = Cin: `w_planes` of conv2d_forward) or, transposed, of its transpose (rows = Cin, cols >= Cout zero padded:
    `wt_planes` of conv2d_dgrad).  "bf16x3": 3 bfloat16 pieces.  "f16x2": 2 float16 pieces (16-bit containers, dtype
    bfloat16 only names the width) of w scaled by the power of two that w_absmax (absmax(w), the block the convolution
    is later given as its weight magnitude) implies."""
    Cout, R, S, Cin = w.shape
    assert w.dtype == torch.float32 and w.is_contiguous()
    npc = plane_pieces(math)
    assert npc == 3 or w_absmax is not None, "f16x2 planes are cut relative to the weight's magnitude block"
    rows, cols = (Cin, cols or (Cout + 31) // 32 * 32) if transposed else (Cout, Cin)
    assert cols % 32 == 0
    if out is None:
        out = empty(rows, R * S, cols // 32, npc, 32, device=w.device, dtype=torch.bfloat16)
    assert out.numel() == rows * R * S * cols * npc and out.dtype == torch.bfloat16
    check(L().dspn_conv2d_weight_planes_f32(ptr(w), 0 if transposed else ptr(out), ptr(out) if transposed else 0, Cout,
                                            R * S, Cin, cols if transposed else 0, npc,
                                            ptr(w_absmax) if npc == 2 else 0, stream()), "weight_planes")
    return out


def weight_planes_table(entries, device):
    """entries: [(w float32 [Cout,R,S,Cin], planes or None, planes_t or None[, w_absmax])] -> (device table, rows, total
    tiles) for weight_planes_batch: every piece-plane operand of a graph refreshed from the float masters by ONE launch.
    With w_absmax (the weight's magnitude block, filled before the launch) the row's planes are the two float16 pieces."""
    import numpy as np
    rows = np.zeros(len(entries), dtype=[("w", "<u8"), ("planes", "<u8"), ("planes_t", "<u8"), ("K", "<i4"), ("T", "<i4"),
                                         ("C", "<i4"), ("cols_t", "<i4"), ("begin", "<i8"), ("absmax", "<u8"),
                                         ("npc", "<i4"), ("pad", "<i4")])
    total = 0
    for i, e in enumerate(entries):
        w, pl, plt = e[:3]
        am = e[3] if len(e) > 3 else None
        npc = 3 if am is None else 2
        Cout, R, S, Cin = w.shape
        assert w.dtype == torch.float32 and w.is_contiguous() and (pl is not None or plt is not None)
        assert pl is None or (pl.dtype == torch.bfloat16 and pl.shape == (Cout, R * S, Cin // 32, npc, 32) and Cin % 32 == 0)
        cols_t = 0
        if plt is not None:
            cols_t = plt.shape[2] * 32
            assert plt.dtype == torch.bfloat16 and plt.shape == (Cin, R * S, cols_t // 32, npc, 32) and cols_t >= Cout
        rows[i] = (w.data_ptr(), 0 if pl is None else pl.data_ptr(), 0 if plt is None else plt.data_ptr(), Cout, R * S, Cin,
                   cols_t, total, 0 if am is None else am.data_ptr(), npc, 0)
        total += int(L().dspn_conv2d_weight_planes_tiles(Cout, R * S, Cin, cols_t, int(plt is not None)))
    assert rows.dtype.itemsize == 64
    return torch.from_numpy(rows.view(np.uint8).copy()).to(device), len(entries), total


def weight_planes_batch(table, n, total):
    check(L().dspn_conv2d_weight_planes_batch_f32(ptr(table), n, total, stream()), "weight_planes_batch")


def conv2d_forward(x, w, bias=None, stride=1, pad=0, dil=1, relu=False, out=None, accumulate=False, residual=None,
                   in_affine=None, out_stats=None, w_planes=None, math=None, x_absmax=None, w_absmax=None,
                   out_minmax=None, x_planes=False, out_absmax=None):
    """x (N,H,W,Cin) ; w (Cout,R,S,Cin) -> (N,Ho,Wo,ldc) with ldc = out.shape[3] if out is given else pad4(Cout).
    in_affine = (scale (Cin,), shift (Cin,), relu): convolve (relu)(x * scale + shift) instead of x.
    math: "fp32" / "bf16" / "bf16x3" / "f16x2" (default: set_conv_math's).  w_planes: weight_planes(w), used in the split
    math modes when Cin % 32 == 0 (made here, one extra launch, when the caller keeps none).
    x_planes: x holds the fp16 piece planes bn_apply_planes wrote (cut by x_absmax), "f16x2" only.
    out_absmax ("f16x2", float tensors, no out_stats, dense output): a 64-float magnitude block that receives the partial
    maxima of |out| as stored (not zeroed here) -- the x_absmax of the next convolution without a pass over out."""
    N, H, W, Cin = x.shape
    Cout, R, S, Cw = w.shape
    assert Cw == Cin, (w.shape, x.shape)
    math = _math_code(math)
    assert not x_planes or (math == 3 and x.dtype == torch.float32 and x_absmax is not None and in_affine is None)
    assert out_minmax is None or (out_stats is not None and out_minmax.numel() == out_stats.numel())
    assert out_absmax is None or (out_stats is None and math == 3 and x.dtype == torch.float32 and out_absmax.numel() == ABSMAX_SLOTS)
    if math == 3 and x.dtype == torch.float32:     # "f16x2": operand magnitudes (made here when the caller keeps none)
        assert w_planes is None or w_absmax is not None, "f16x2 planes come with the magnitude block they were cut by"
        x_absmax = absmax(x, in_affine) if x_absmax is None else x_absmax
        w_absmax = absmax(w) if w_absmax is None else w_absmax
    else:
        x_absmax = w_absmax = None
    if needs_planes(x.dtype, Cin, math):
        if w_planes is None:
            w_planes = weight_planes(w, math=math, w_absmax=w_absmax)
    else:
        w_planes = None
    ph, pw = _hw(pad)
    Ho, Wo = conv_out_size(H, R, stride, ph, dil), conv_out_size(W, S, stride, pw, dil)
    if out is None:
        ldc = padc(Cout, x.dtype)
        out = (zeros if ldc != Cout else empty)(N, Ho, Wo, ldc, device=x.device, dtype=x.dtype)
    ldc = out.shape[3]
    assert w.dtype == x.dtype == out.dtype, (x.dtype, w.dtype, out.dtype)
    ws = workspace(L().dspn_conv2d_split_workspace_bytes(N * Ho * Wo, Cout), x.device, "split")
    assert residual is None or (residual.shape == out.shape and residual.dtype == out.dtype)
    sc, sh, arelu = in_affine if in_affine is not None else (None, None, False)
    check(_f("dspn_conv2d_forward_bn", x)(ptr(x), ptr(sc), ptr(sh), int(arelu), ptr(w), ptr(w_planes), ptr(bias),
                                         ptr(residual), ptr(out),
                                         N, H, W, Cin, Cout, R, S, stride, ph, pw, dil, Ho, Wo, 0, ldc, int(relu),
                                         int(accumulate), ptr(out_stats), 0 if out_stats is None else out_stats.numel() * 4,
                                         ptr(out_minmax if out_absmax is None else out_absmax),      # (without out_stats: the magnitude block)
                                         math | (MATH_X_PLANES if x_planes else 0), ptr(x_absmax),
                                         ptr(w_absmax), ptr(ws), ws.numel(), stream()),
          "conv2d_forward")
    return out


def weight_transpose_table(triples, device):
    """triples: [(w float32 [Cout,R,S,Cin], wt [Cin,R,S,Kp], wh or None)] -> (device table, rows, total tiles, bf16?)
    for weight_transpose_batch.  float32 operands: wt float32, wh None.  bfloat16 operands: wt bfloat16 and wh = the
    bfloat16 copy of w itself (the forward operand), both refreshed from the float32 master by the one launch."""
    import numpy as np
    rows = np.zeros(len(triples), dtype=[("w", "<u8"), ("wt", "<u8"), ("K", "<i4"), ("T", "<i4"), ("C", "<i4"),
                                         ("Kp", "<i4"), ("begin", "<i8"), ("wh", "<u8")])
    total = 0
    half = triples[0][1].dtype == torch.bfloat16
    for i, (w, wt, wh) in enumerate(triples):
        Cout, R, S, Cin = w.shape
        assert w.dtype == torch.float32 and wt.shape[:3] == (Cin, R, S) and w.is_contiguous() and wt.is_contiguous()
        assert (wt.dtype == torch.bfloat16) == half and (wh is None or (half and wh.shape == w.shape))
        rows[i] = (w.data_ptr(), wt.data_ptr(), Cout, R * S, Cin, wt.shape[3], total, 0 if wh is None else wh.data_ptr())
        total += int(L().dspn_conv2d_weight_transpose_tiles(Cout, R * S, Cin, wt.shape[3]))     # 32 x 32 tiles per tap
    assert rows.dtype.itemsize == 48
    table = torch.from_numpy(rows.view(np.uint8).copy()).to(device)
    return table, len(triples), total, half


def weight_transpose_batch(table, n, total, half=False):
    f = L().dspn_conv2d_weight_prepare_batch_bf16 if half else L().dspn_conv2d_weight_transpose_batch_f32
    check(f(ptr(table), n, total, stream()), "weight_transpose_batch")


def weight_transpose(w, out=None, dtype=None, copy=None):
    """float32 master [Cout,R,S,Cin] -> [Cin,R,S,padc(Cout)] (zero padded), the operand of conv2d_dgrad, in `dtype`
    (default: out's, else float32); bfloat16: `copy` (optional, w's shape) also receives the bf16 copy of w itself"""
    Cout, R, S, Cin = w.shape
    dtype = out.dtype if out is not None else (dtype or torch.float32)
    Kp = padc(Cout, dtype) if out is None else out.shape[3]
    if out is None:
        out = empty(Cin, R, S, Kp, device=w.device, dtype=dtype)
    if dtype == torch.bfloat16:
        check(L().dspn_conv2d_weight_prepare_bf16(ptr(w), ptr(copy), ptr(out), Cout, R * S, Cin, Kp, stream()),
              "weight_prepare")
    else:
        check(L().dspn_conv2d_weight_transpose_f32(ptr(w), ptr(out), Cout, R * S, Cin, Kp, stream()),
              "weight_transpose")
    return out


def weight_cast(w):
    """bf16 operand copy of a float32 master weight (one-off helper for tests / small graphs)"""
    return w.to(torch.bfloat16)


def conv_dgrad_bn_tiles(x_shape, stride):
    N, H, W, C = x_shape
    return L().dspn_conv2d_dgrad_bn_tiles(N, H, W, C, stride)


MATH_DY_PLANES = 0x200       # include/dspn_nn.h DSPN_MATH_DY_PLANES
MATH_X_PLANES = 0x400        # include/dspn_nn.h DSPN_MATH_X_PLANES


def conv2d_dgrad(dy, wt, x_shape, stride=1, pad=0, dil=1, out=None, accumulate=False, bn_bwd=None, wt_planes=None,
                 math=None, dy_absmax=None, w_absmax=None, bn_dy_absmax=None, dy_planes=False, wt_shape=None):
    """dy (N,Ho,Wo,ldy), wt (Cin,R,S,ldy) -> dx (N,H,W,ldc>=Cin).
    bn_bwd = (bn_x, scale, shift, mean, rstd, relu, sums): dx is the complete gradient of a BatchNorm(+ReLU) output
    whose input was bn_x; the two reductions of its backward pass are written to sums (tiles, 2, Cin).
    wt_planes: the piece planes of wt (weight_planes(w, transposed=True, cols=ldy)), used in the split math when
    ldy % 32 == 0 (made here from wt when the caller keeps none)."""
    N, H, W, Cx = x_shape
    # wt may be None when the call reads piece planes (wt_planes, split math with ldy % 32 == 0): wt_shape = (Cin, R, S, ldy)
    Cin, R, S, ldy = wt.shape if wt is not None else wt_shape
    assert dy.shape[3] == ldy, (dy.shape, (Cin, R, S, ldy))
    assert wt is not None or wt_planes is not None
    math = _math_code(math)
    if math == 3 and dy.dtype == torch.float32:
        assert wt_planes is None or w_absmax is not None, "f16x2 planes come with the magnitude block they were cut by"
        assert not dy_planes or dy_absmax is not None, "piece-plane gradients come with the magnitude block they were cut by"
        dy_absmax = absmax(dy) if dy_absmax is None else dy_absmax
        w_absmax = absmax(wt) if w_absmax is None else w_absmax
    else:
        dy_absmax = w_absmax = None
    if needs_planes(dy.dtype, ldy, math):
        if wt_planes is None:                    # planes of the matrix wt itself: rows = Cin, cols = ldy
            wt_planes = weight_planes(wt, math=math, w_absmax=w_absmax)
    else:
        wt_planes = None
    Ho, Wo = dy.shape[1], dy.shape[2]
    if out is None:
        out = (zeros if Cx != Cin else empty)(N, H, W, Cx, device=dy.device, dtype=dy.dtype)
    assert (wt is None or wt.dtype == dy.dtype) and dy.dtype == out.dtype, (dy.dtype, out.dtype)
    ws = workspace(L().dspn_conv2d_split_workspace_bytes(N * H * W, Cin), dy.device, "split")
    ph, pw = _hw(pad)
    bx, bsc, bsh, bmu, brs, brelu, bsums = bn_bwd if bn_bwd is not None else (None, None, None, None, None, False, None)
    check(_f("dspn_conv2d_dgrad_bn", dy)(ptr(dy), ptr(wt), ptr(wt_planes), ptr(out), N, H, W, Cin, ldy, R, S, stride, ph, pw,
                                       dil, Ho, Wo, out.shape[3], int(accumulate), ptr(bx), ptr(bsc), ptr(bsh), ptr(bmu),
                                       ptr(brs), int(brelu), ptr(bsums), 0 if bsums is None else bsums.numel() * 4,
                                       ptr(bn_dy_absmax), math | (MATH_DY_PLANES if dy_planes else 0), ptr(dy_absmax),
                                       ptr(w_absmax), ptr(ws), ws.numel(), stream()), "conv2d_dgrad")
    return out


def _wgrad_absmax(x, dy, in_affine, math, x_absmax, dy_absmax):
    if math == 3 and x.dtype == torch.float32:      # (piece-plane gradients always come with their block: conv2d_wgrad's assert)
        return (absmax(x, in_affine) if x_absmax is None else x_absmax, absmax(dy) if dy_absmax is None else dy_absmax)
    return None, None


def conv2d_wgrad(x, dy, w_shape, stride=1, pad=0, dil=1, out=None, accumulate=False, in_affine=None, math=None,
                 x_absmax=None, dy_absmax=None, dy_planes=False, x_planes=False):
    """x (N,H,W,Cin), dy (N,Ho,Wo,ldy) -> dw (Cout,R,S,Cin); in_affine as in conv2d_forward; x_planes / dy_planes: the
    operand is fp16 piece planes cut by its magnitude block ("f16x2")"""
    N, H, W, Cin = x.shape
    Cout, R, S, Cw = w_shape
    assert Cw == Cin
    Ho, Wo, ldy = dy.shape[1], dy.shape[2], dy.shape[3]
    if out is None:
        out = empty(Cout, R, S, Cin, device=x.device)
    nbytes = L().dspn_conv2d_wgrad_workspace_bytes(N, Ho, Wo, Cin, Cout, R, S)
    ws = workspace(nbytes, x.device, "wgrad")
    ph, pw = _hw(pad)
    sc, sh, arelu = in_affine if in_affine is not None else (None, None, False)
    assert x.dtype == dy.dtype and out.dtype == torch.float32
    math = _math_code(math)
    assert not x_planes or (x_absmax is not None and in_affine is None)
    x_absmax, dy_absmax = _wgrad_absmax(x, dy, in_affine, math, x_absmax, dy_absmax)
    check(_f("dspn_conv2d_wgrad_bn", x)(ptr(x), ptr(sc), ptr(sh), int(arelu), ptr(dy), ptr(out), N, H, W, Cin, Cout, ldy,
                                       R, S, stride, ph, pw, dil, Ho, Wo, int(accumulate),
                                       math | (MATH_DY_PLANES if dy_planes else 0) | (MATH_X_PLANES if x_planes else 0),
                                       ptr(x_absmax), ptr(dy_absmax), ptr(ws), ws.numel(), stream()), "conv2d_wgrad")
    return out


def tap_sum(z, bias, cout, R, S, pad, out):
    """z (N,H,W,ldz >= cout*R*S) -> out (N,H,W,ldy): shifted sum over taps (+ bias); pad channels of out zeroed"""
    N, H, W, ldz = z.shape
    ph, pw = _hw(pad)
    assert z.dtype == out.dtype
    check(_f("dspn_tap_sum", z)(ptr(z), ptr(bias), ptr(out), N, H, W, cout, out.shape[3], ldz, R, S, ph, pw, stream()),
          "tap_sum")
    return out


def tap_spread(dy, cout, R, S, pad, out):
    """dy (N,H,W,ldy) -> out (N,H,W,ldz >= cout*R*S): gradient of tap_sum with respect to z"""
    N, H, W, ldy = dy.shape
    ph, pw = _hw(pad)
    assert dy.dtype == out.dtype
    check(_f("dspn_tap_spread", dy)(ptr(dy), ptr(out), N, H, W, cout, ldy, out.shape[3], R, S, ph, pw, stream()),
          "tap_spread")
    return out


def conv2d_wgrad_splits(x_shape, dy_shape, w_shape, stride):
    N, H, W, Cin = x_shape
    Cout, R, S, _ = w_shape
    return L().dspn_conv2d_wgrad_splits(N, dy_shape[1], dy_shape[2], Cin, Cout, R, S, stride)


def conv2d_wgrad_slabs(x, dy, w_shape, slabs, stride=1, pad=0, dil=1, in_affine=None, math=None, x_absmax=None,
                       dy_absmax=None, dy_planes=False, x_planes=False):
    """the weight-gradient GEMM alone: split-K partial sums -> slabs (splits, Cout, R, S, Cin); see slab_reduce_batch"""
    N, H, W, Cin = x.shape
    Cout, R, S, Cw = w_shape
    assert Cw == Cin
    ph, pw = _hw(pad)
    sc, sh, arelu = in_affine if in_affine is not None else (None, None, False)
    assert x.dtype == dy.dtype and slabs.dtype == torch.float32
    math = _math_code(math)
    assert not x_planes or (x_absmax is not None and in_affine is None)
    x_absmax, dy_absmax = _wgrad_absmax(x, dy, in_affine, math, x_absmax, dy_absmax)
    check(_f("dspn_conv2d_wgrad_slabs", x)(ptr(x), ptr(sc), ptr(sh), int(arelu), ptr(dy), ptr(slabs), slabs.numel() * 4,
                                          N, H, W, Cin, Cout, dy.shape[3], R, S, stride, ph, pw, dil, dy.shape[1],
                                          dy.shape[2],
                                          math | (MATH_DY_PLANES if dy_planes else 0) | (MATH_X_PLANES if x_planes else 0),
                                          ptr(x_absmax), ptr(dy_absmax), stream()), "conv2d_wgrad_slabs")


def slab_reduce_table(entries, device):
    """entries: [(slabs (splits, ...), dw, accumulate)] -> (device table, rows, total float4)"""
    import numpy as np
    rows = np.zeros(len(entries), dtype=[("slab", "<u8"), ("dw", "<u8"), ("n4", "<i8"), ("splits", "<i4"),
                                         ("acc", "<i4"), ("begin", "<i8")])
    total = 0
    for i, (slabs, dw, acc) in enumerate(entries):
        assert dw.numel() % 4 == 0 and slabs.numel() == slabs.shape[0] * dw.numel()
        rows[i] = (slabs.data_ptr(), dw.data_ptr(), dw.numel() // 4, slabs.shape[0], int(acc), total)
        total += dw.numel() // 4
    assert rows.dtype.itemsize == 40
    return torch.from_numpy(rows.view(np.uint8).copy()).to(device), len(entries), total


def slab_reduce_batch(table, n, total4):
    check(L().dspn_conv2d_slab_reduce_batch_f32(ptr(table), n, total4, stream()), "slab_reduce_batch")


def conv2d_input_sum_grad(dy, w, x_shape, stride=1, pad=0, dil=1, out=None):
    """sum over all pixels of the conv's data gradient, per input channel: (Cin,) without forming dx"""
    N, H, W, Cin = x_shape
    Cout, R, S, _ = w.shape
    Ho, Wo, ldy = dy.shape[1], dy.shape[2], dy.shape[3]
    out = empty(Cin, device=dy.device) if out is None else out
    ws = workspace(L().dspn_conv2d_input_sum_grad_workspace_bytes(Ho, Wo, ldy, R, S), dy.device, "sumgrad")
    ph, pw = _hw(pad)
    check(_f("dspn_conv2d_input_sum_grad", dy)(ptr(dy), ptr(w), ptr(out), N, H, W, Cin, Cout, ldy, R, S, stride,
                                             ph, pw, dil, Ho, Wo, ptr(ws), ws.numel(), stream()),
          "conv2d_input_sum_grad")
    return out


# ------------------------------------------------------------------ batch norm
def _rows(x):
    return x.numel() // x.shape[-1]


def bn_stats(x, eps, gamma, beta, mean=None, rstd=None, scale=None, shift=None):
    """batch statistics + folded affine; returns (mean, rstd, scale, shift)"""
    C = x.shape[-1]
    rows = _rows(x)
    mean = empty(C, device=x.device) if mean is None else mean
    rstd = empty(C, device=x.device) if rstd is None else rstd
    scale = empty(C, device=x.device) if scale is None else scale
    shift = empty(C, device=x.device) if shift is None else shift
    ws = workspace(L().dspn_bn_workspace_bytes(rows, C), x.device, "bn")
    check(_f("dspn_bn_stats", x)(ptr(x), rows, C, eps, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(scale),
                                ptr(shift), ptr(ws), ws.numel(), stream()), "bn_stats")
    return mean, rstd, scale, shift


def bn_apply(x, scale, shift, relu=False, out=None, out_absmax=None):
    """out_absmax (float tensors): a 64-float magnitude block that receives the partial maxima of |out| (not zeroed here)"""
    out = torch.empty_like(x) if out is None else out
    assert out.dtype == x.dtype and (out_absmax is None or out_absmax.numel() == ABSMAX_SLOTS)
    check(_f("dspn_bn_apply", x)(ptr(x), ptr(scale), ptr(shift), ptr(out), _rows(x), x.shape[-1], int(relu),
                                ptr(out_absmax if x.dtype == torch.float32 else None), stream()), "bn_apply")
    return out


def bn_apply_planes(x, scale, shift, y_absmax, relu=False, out=None):
    """(relu)(x * scale + shift) as fp16 piece planes [pixel][C / 32][piece][32] in a buffer of x's shape and bytes, cut by
    the magnitude block y_absmax (known beforehand: bn_stats_from_tiles out_absmax) -- the x operand of conv2d_forward /
    conv2d_wgrad(x_planes=True)"""
    out = torch.empty_like(x) if out is None else out
    assert x.dtype == out.dtype == torch.float32 and x.shape[-1] % 32 == 0 and y_absmax.numel() == ABSMAX_SLOTS
    check(L().dspn_bn_apply_planes_f32(ptr(x), ptr(scale), ptr(shift), ptr(out), _rows(x), x.shape[-1], int(relu),
                                       ptr(y_absmax), stream()), "bn_apply_planes")
    return out


def absmax_affine_bound(scale, shift, x_absmax, out):
    """out (64-float magnitude block, not zeroed here) max= max_c |scale[c]| * M + |shift[c]|, M = the magnitude in x_absmax:
    a bound of |(relu)(x * scale + shift)| without a pass over x"""
    assert scale.numel() == shift.numel() and x_absmax.numel() == out.numel() == ABSMAX_SLOTS
    check(L().dspn_absmax_affine_bound_f32(ptr(scale), ptr(shift), scale.numel(), ptr(x_absmax), ptr(out), stream()),
          "absmax_affine_bound")
    return out


def bn_backward(x, scale, shift, dy, mean, rstd, gamma, relu=False, dx=None, dgamma=None, dbeta=None,
                accumulate=False, dx_absmax=None):
    """dx_absmax: 64-float magnitude block that receives the partial maxima of |dx| as stored (see absmax)"""
    C = x.shape[-1]
    rows = _rows(x)
    dx = torch.empty_like(x) if dx is None else dx
    dbeta = empty(C, device=x.device) if dbeta is None else dbeta
    if gamma is not None and dgamma is None:
        dgamma = empty(C, device=x.device)
    ws = workspace(L().dspn_bn_workspace_bytes(rows, C), x.device, "bn")
    assert dy.dtype == x.dtype == dx.dtype
    check(_f("dspn_bn_backward", x)(ptr(x), ptr(scale), ptr(shift), ptr(dy), ptr(mean), ptr(rstd), ptr(gamma), ptr(dx),
                                   ptr(dgamma), ptr(dbeta), rows, C, int(relu), int(accumulate), ptr(dx_absmax), ptr(ws),
                                   ws.numel(), stream()), "bn_backward")
    return dx, dgamma, dbeta


def bn_backward_maxpool(x, scale, shift, dy_pool, argmax, k, stride, pad, mean, rstd, gamma, relu=False, dx=None, dgamma=None,
                        dbeta=None, dx_absmax=None):
    """bn_backward whose output gradient is the max-pooling backward of dy_pool (argmax record), formed on the fly: the
    BatchNorm(+ReLU) -> max pooling pair of the resnet stem without the dense gradient tensor in between (float32)"""
    N, H, W, C = x.shape
    assert x.dtype == dy_pool.dtype == torch.float32 and argmax.dtype == torch.uint8 and argmax.shape == dy_pool.shape
    dx = torch.empty_like(x) if dx is None else dx
    dbeta = empty(C, device=x.device) if dbeta is None else dbeta
    if gamma is not None and dgamma is None:
        dgamma = empty(C, device=x.device)
    ws = workspace(L().dspn_bn_workspace_bytes(N * H * W, C), x.device, "bn")
    check(L().dspn_bn_backward_maxpool_f32(ptr(x), ptr(scale), ptr(shift), ptr(dy_pool), ptr(argmax), N, H, W, C, k, stride, pad,
                                           dy_pool.shape[1], dy_pool.shape[2], ptr(mean), ptr(rstd), ptr(gamma), ptr(dx),
                                           ptr(dgamma), ptr(dbeta), int(relu), ptr(dx_absmax), ptr(ws), ws.numel(), stream()),
          "bn_backward_maxpool")
    return dx, dgamma, dbeta


def bn_backward_from_sums(x, scale, shift, dy, mean, rstd, gamma, sums, tiles, relu=False, dx=None, dgamma=None,
                          dbeta=None, accumulate=False, dx_absmax=None, dy_absmax=None, x_chan_minmax=None, dx_planes=False,
                          dx_absmin=None, phase=0, park=False):
    """bn_backward with the two reductions already gathered per row tile (conv2d_dgrad(bn_bwd=...)).
    dx_planes ("f16x2" math): dx is written as fp16 piece planes (same bytes, same buffer shape) cut by the power of two of a
    BOUND of |dx| that is formed from dy_absmax (the magnitude block of dy, conv2d_dgrad's bn_dy_absmax) and x_chan_minmax
    (2 x C per-channel extremes of x, bn_stats_from_tiles' out_chan_minmax) and left in dx_absmax.
    phase (round 6): 1 = the finalize alone (coefficients into the "bn" workspace of the current lane), 2 = the apply pass alone
    from what a phase-1 call with the same arguments left there; 0 = both.  park (with phase 1): the finalize is not launched but
    parked for the NEXT weight-gradient launch on this stream, in front of whose grid it rides (csrc/bn_final_job.h); the
    phase-2 call runs it on its own if no weight gradient came by"""
    assert phase in (0, 1, 2) and (not park or phase == 1)
    C = x.shape[-1]
    rows = _rows(x)
    dx = torch.empty_like(x) if dx is None else dx
    dbeta = empty(C, device=x.device) if dbeta is None else dbeta
    if gamma is not None and dgamma is None:
        dgamma = empty(C, device=x.device)
    ws = workspace(12 * C + L().dspn_bn_tiles_workspace_bytes(tiles, C), x.device, "bn")
    assert dy.dtype == x.dtype == dx.dtype
    check(_f("dspn_bn_backward_from_sums", x)(ptr(x), ptr(scale), ptr(shift), ptr(dy), ptr(mean), ptr(rstd), ptr(gamma),
                                             ptr(sums), tiles, ptr(dx), ptr(dgamma), ptr(dbeta), rows, C, int(relu),
                                             int(accumulate), ptr(dx_absmax), ptr(dx_absmin), ptr(dy_absmax), ptr(x_chan_minmax),
                                             int(bool(dx_planes)) | (phase << 1) | (8 if park else 0), ptr(ws), ws.numel(), stream()),
          "bn_backward_from_sums")
    return dx, dgamma, dbeta


# ------------------------------------------------------------------ element-wise / layout
def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    assert a.dtype == b.dtype == out.dtype
    check(_f("dspn_add", a)(ptr(a), ptr(b), ptr(out), a.numel(), stream()), "add")
    return out


def relu_backward(y, dy, dx=None, accumulate=False):
    dx = torch.empty_like(y) if dx is None else dx
    assert y.dtype == dy.dtype == dx.dtype
    check(_f("dspn_relu_backward", y)(ptr(y), ptr(dy), ptr(dx), y.numel(), int(accumulate), stream()),
          "relu_backward")
    return dx


def relu_backward_colsum(y, dy, C, dx=None, out=None, dx_absmax=None):
    """dx = (y > 0) * dy (in place by default) and the column sums of dx over all rows, in one pass; dx_absmax (float
    tensors): a 64-float magnitude block that receives the partial maxima of |dx| (not zeroed here)"""
    dx = dy if dx is None else dx
    ld = y.shape[-1]
    rows = _rows(y)
    out = empty(C, device=y.device) if out is None else out
    ws = workspace(L().dspn_colsum_workspace_bytes(rows, C), y.device, "colsum")
    assert y.dtype == dy.dtype == dx.dtype
    assert dx_absmax is None or (y.dtype == torch.float32 and dx_absmax.numel() == ABSMAX_SLOTS)
    check(_f("dspn_relu_backward_colsum", y)(ptr(y), ptr(dy), ptr(dx), rows, C, ld, ptr(out), ptr(dx_absmax), ptr(ws), ws.numel(),
                                             stream()), "relu_backward_colsum")
    return dx, out


def fill(t, v):
    check(L().dspn_fill_f32(ptr(t), float(v), t.numel(), stream()), "fill")
    return t


def colsum(a, C, out=None):
    """sum over all rows of the first C channels of a (..., ld) tensor"""
    ld = a.shape[-1]
    rows = _rows(a)
    out = empty(C, device=a.device) if out is None else out
    ws = workspace(L().dspn_colsum_workspace_bytes(rows, C), a.device, "colsum")
    check(_f("dspn_colsum", a)(ptr(a), rows, C, ld, ptr(out), ptr(ws), ws.numel(), stream()), "colsum")
    return out


def nchw_to_nhwc(src, Cp=None, out=None):
    """float32 NCHW -> NHWC in out's storage type (float32 / bfloat16), channels zero padded to Cp"""
    N, C, H, W = src.shape
    Cp = (pad4(C) if out is None else out.shape[3]) if Cp is None else Cp
    out = empty(N, H, W, Cp, device=src.device) if out is None else out
    assert src.dtype == torch.float32
    check(_f("dspn_nchw_to_nhwc", out)(ptr(src), ptr(out), N, C, H, W, Cp, stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(src, C=None, out=None):
    N, H, W, Cp = src.shape
    C = Cp if C is None else C
    out = empty(N, C, H, W, device=src.device) if out is None else out
    check(L().dspn_nhwc_to_nchw_f32(ptr(src), ptr(out), N, C, H, W, Cp, stream()), "nhwc_to_nchw")
    return out


def copy_block_table(entries, device):
    """entries: [(src, dst, samples, rows_per_sample, C, src_sample_stride, lds, soff, dst_sample_stride, ldd, doff, accumulate)]
    (the arguments of copy_block, float32 tensors) -> (device table, rows, total elements) for copy_block_batch"""
    import numpy as np
    rows = np.zeros(len(entries), dtype=[("src", "<u8"), ("dst", "<u8"), ("rps", "<i8"), ("sss", "<i8"), ("dss", "<i8"),
                                         ("C", "<i4"), ("lds", "<i4"), ("soff", "<i4"), ("ldd", "<i4"), ("doff", "<i4"),
                                         ("acc", "<i4"), ("begin", "<i8")])
    assert rows.dtype.itemsize == 72
    total = 0
    for i, (src, dst, samples, rps, C, sss, lds, soff, dss, ldd, doff, acc) in enumerate(entries):
        assert src.dtype == dst.dtype == torch.float32 and samples > 0 and rps > 0 and C > 0
        rows[i] = (src.data_ptr(), dst.data_ptr(), rps, sss, dss, C, lds, soff, ldd, doff, int(bool(acc)), total)
        total += samples * rps * C
    return torch.from_numpy(rows.view(np.uint8).copy()).to(device), len(entries), total


def copy_block_batch(table, n, total):
    check(L().dspn_copy_block_batch_f32(ptr(table), n, total, stream()), "copy_block_batch")


def copy_block(src, dst, samples, rows_per_sample, C, src_sample_stride, lds, soff, dst_sample_stride, ldd,
               doff, accumulate=False):
    """strided block copy; src / dst may differ in storage type (bf16 map -> float loss input and back)"""
    if src.dtype == dst.dtype:
        f = _f("dspn_copy_block", src)
    elif src.dtype == torch.bfloat16:
        f = L().dspn_copy_block_bf16_f32
    else:
        f = L().dspn_copy_block_f32_bf16
    check(f(ptr(src), ptr(dst), samples, rows_per_sample, C, src_sample_stride, lds,
                                  soff, dst_sample_stride, ldd, doff, int(accumulate), stream()), "copy_block")
    return dst


def transpose_bnc(src, out=None):
    B, N, C = src.shape
    out = empty(B, C, N, device=src.device) if out is None else out
    check(L().dspn_transpose_bnc_f32(ptr(src), ptr(out), B, N, C, stream()), "transpose_bnc")
    return out


# ------------------------------------------------------------------ pooling / sampler
def maxpool_forward(x, k, stride, pad, out=None, argmax=None, in_affine=None, out_absmax=None):
    """argmax: optional uint8 tensor of the output's shape receiving the window position of each maximum.
    in_affine = (scale, shift, relu): pool (relu)(x * scale + shift) instead of x (the BatchNorm in front folded in);
    out_absmax then optionally receives the magnitude block of the pooled output (float tensors)"""
    N, H, W, C = x.shape
    if out is None:
        out = empty(N, conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad), C, device=x.device, dtype=x.dtype)
    Ho, Wo = out.shape[1], out.shape[2]     # a larger (pooling_convention='full') output is the caller's choice
    assert argmax is None or (argmax.dtype == torch.uint8 and argmax.shape == out.shape)
    assert out.dtype == x.dtype
    if in_affine is not None:
        sc, sh, relu = in_affine
        check(_f("dspn_maxpool_forward_bn", x)(ptr(x), ptr(sc), ptr(sh), int(bool(relu)), ptr(out), ptr(argmax), N, H, W, C, k,
                                             stride, pad, Ho, Wo, ptr(out_absmax if x.dtype == torch.float32 else None), stream()),
              "maxpool_forward_bn")
        return out
    check(_f("dspn_maxpool_forward", x)(ptr(x), ptr(out), ptr(argmax), N, H, W, C, k, stride, pad, Ho, Wo, stream()),
          "maxpool_forward")
    return out


def maxpool_backward_argmax(argmax, dy, x_shape, k, stride, pad, dx=None):
    N, H, W, C = x_shape
    dx = empty(N, H, W, C, device=dy.device, dtype=dy.dtype) if dx is None else dx
    assert dx.dtype == dy.dtype
    check(_f("dspn_maxpool_backward_argmax", dy)(ptr(argmax), ptr(dy), ptr(dx), N, H, W, C, k, stride, pad,
                                               dy.shape[1], dy.shape[2], stream()), "maxpool_backward_argmax")
    return dx


def maxpool_backward(x, y, dy, k, stride, pad, dx=None):
    N, H, W, C = x.shape
    dx = torch.empty_like(x) if dx is None else dx
    assert x.dtype == y.dtype == dy.dtype == dx.dtype
    check(_f("dspn_maxpool_backward", x)(ptr(x), ptr(y), ptr(dy), ptr(dx), N, H, W, C, k, stride, pad,
                                        y.shape[1], y.shape[2], stream()), "maxpool_backward")
    return dx


def avgpool_forward(x, k, out=None):
    N, H, W, C = x.shape
    Ho, Wo = H // k, W // k
    out = empty(N, Ho, Wo, C, device=x.device, dtype=x.dtype) if out is None else out
    assert out.dtype == x.dtype
    check(_f("dspn_avgpool_forward", x)(ptr(x), ptr(out), N, H, W, C, k, Ho, Wo, stream()), "avgpool_forward")
    return out


def avgpool2d_forward(x, k, stride, pad, out=None):
    """overlapping average pooling, divisor k*k (padding counted)"""
    N, H, W, C = x.shape
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    out = empty(N, Ho, Wo, C, device=x.device, dtype=x.dtype) if out is None else out
    assert out.dtype == x.dtype
    check(_f("dspn_avgpool2d_forward", x)(ptr(x), ptr(out), N, H, W, C, k, stride, pad, Ho, Wo, stream()),
          "avgpool2d_forward")
    return out


def avgpool2d_backward(dy, x_shape, k, stride, pad, dx=None, accumulate=False):
    N, H, W, C = x_shape
    dx = empty(N, H, W, C, device=dy.device, dtype=dy.dtype) if dx is None else dx
    assert dx.dtype == dy.dtype
    check(_f("dspn_avgpool2d_backward", dy)(ptr(dy), ptr(dx), N, H, W, C, k, stride, pad, dy.shape[1], dy.shape[2],
                                          int(accumulate), stream()), "avgpool2d_backward")
    return dx


def avgpool_backward(dy, x_shape, k, dx=None, accumulate=False):
    N, H, W, C = x_shape
    dx = empty(N, H, W, C, device=dy.device, dtype=dy.dtype) if dx is None else dx
    assert dx.dtype == dy.dtype
    check(_f("dspn_avgpool_backward", dy)(ptr(dy), ptr(dx), N, H, W, C, k, dy.shape[1], dy.shape[2],
                                        int(accumulate), stream()), "avgpool_backward")
    return dx


def bilinear_forward(x, out, coff, accumulate=False):
    """resize x (N,Hin,Win,C) into (accumulate: onto) channels [coff, coff+C) of out (N,Ho,Wo,ldo)"""
    N, Hin, Win, C = x.shape
    f = L().dspn_bilinear_forward_acc_f32 if accumulate else L().dspn_bilinear_forward_f32
    check(f(ptr(x), ptr(out), N, Hin, Win, C, out.shape[1], out.shape[2], out.shape[3], coff, stream()),
          "bilinear_forward")
    return out


def bilinear_backward(dy, x_shape, coff, dx=None, separable=True):
    """separable: two passes (along W, then along H) through a scratch tensor; else the one-pass gather kernel"""
    N, Hin, Win, C = x_shape
    dx = empty(N, Hin, Win, C, device=dy.device) if dx is None else dx
    if separable and (Hin, Win) != (dy.shape[1], dy.shape[2]):
        ws = workspace(L().dspn_bilinear_backward_workspace_bytes(N, Win, C, dy.shape[1]), dy.device, "bilinear")
        check(L().dspn_bilinear_backward_ws_f32(ptr(dy), ptr(dx), N, Hin, Win, C, dy.shape[1], dy.shape[2],
                                                dy.shape[3], coff, ptr(ws), ws.numel(), stream()), "bilinear_backward")
        return dx
    check(L().dspn_bilinear_backward_f32(ptr(dy), ptr(dx), N, Hin, Win, C, dy.shape[1], dy.shape[2],
                                         dy.shape[3], coff, stream()), "bilinear_backward")
    return dx


class SamplerSources:
    """host-side table of the source maps of one affine-sampler call: [(x (N,Hin,Win,C), channel offset)]"""

    def __init__(self, entries):
        n = len(entries)
        self.n = n
        self.tensors = [x for x, _ in entries]                  # keep the buffers alive
        self.x = (_c.c_void_p * n)(*[x.data_ptr() for x, _ in entries])
        self.Hin = (_c.c_int * n)(*[x.shape[1] for x, _ in entries])
        self.Win = (_c.c_int * n)(*[x.shape[2] for x, _ in entries])
        self.C = (_c.c_int * n)(*[x.shape[3] for x, _ in entries])
        self.coff = (_c.c_int * n)(*[int(o) for _, o in entries])


def affine_sampler_forward(sources, theta, out):
    """GridGenerator(affine theta, target out.shape[1:3]) + BilinearSampler of every source into its channel slice
    of out (N,Ho,Wo,ldo); coinciding slices are summed, uncovered channels zeroed"""
    N, Ho, Wo, ldo = out.shape
    assert all(t.dtype == out.dtype for t in sources.tensors)
    check(_f("dspn_affine_sampler_forward", out)(sources.x, sources.Hin, sources.Win, sources.C, sources.coff, sources.n,
                                              ptr(theta), ptr(out), N, Ho, Wo, ldo, stream()), "affine_sampler_forward")
    return out


def affine_sampler_backward_data(dy, theta, x_shape, coff, dx=None, accumulate=False):
    N, Hin, Win, C = x_shape
    dx = empty(N, Hin, Win, C, device=dy.device, dtype=dy.dtype) if dx is None else dx
    assert dx.dtype == dy.dtype
    check(_f("dspn_affine_sampler_backward_data", dy)(ptr(dy), ptr(theta), ptr(dx), N, Hin, Win, C, dy.shape[1], dy.shape[2],
                                                    dy.shape[3], coff, int(accumulate), stream()),
          "affine_sampler_backward_data")
    return dx


def affine_sampler_backward_data_theta(dy, theta, x, coff, theta_partial, dx=None, accumulate=False, dx_absmax=None):
    """affine_sampler_backward_data plus this source's rows of the theta gradient from the same pass: x = the source map
    the forward call sampled (it may be the buffer dx when dx is overwritten), theta_partial = a float64 tensor
    (N * Hin * Win, 6) that receives one row per source pixel (affine_sampler_theta_reduce sums the rows of all sources)"""
    N, Hin, Win, C = x.shape
    dx = empty(N, Hin, Win, C, device=dy.device, dtype=dy.dtype) if dx is None else dx
    assert dx.dtype == dy.dtype == x.dtype and theta_partial.dtype == torch.float64 and theta_partial.is_contiguous()
    assert theta_partial.numel() >= 6 * affine_sampler_theta_rows(x.shape, dy.shape[1], x.dtype)
    ws = workspace(L().dspn_affine_sampler_backward_workspace_bytes(N, Hin, Win, C, dy.shape[1]), dy.device, "sampler")
    check(_f("dspn_affine_sampler_backward_data_theta", dy)(ptr(dy), ptr(theta), ptr(x), ptr(dx), N, Hin, Win, C, dy.shape[1],
                                                          dy.shape[2], dy.shape[3], coff, int(accumulate), ptr(theta_partial),
                                                          theta_partial.numel() * 8,
                                                          ptr(dx_absmax if dy.dtype == torch.float32 else None),
                                                          ptr(ws), ws.numel(), stream()),
          "affine_sampler_backward_data_theta")
    return dx


def affine_sampler_theta_rows(x_shape, Ho, dtype=torch.float32):
    """rows of the float64 (rows, 6) tensor affine_sampler_backward_data_theta fills for a source map of shape x_shape sampled
    to Ho target rows (small float maps are split over several workgroups per pixel: one row each)"""
    N, Hin, Win, _ = x_shape
    return N * Hin * Win if dtype != torch.float32 else int(L().dspn_affine_sampler_theta_rows(N, Hin, Win, Ho))


def affine_sampler_theta_reduce(theta_partial, dtheta, accumulate=False):
    """dtheta[6] (+)= fixed-order sum of the rows of theta_partial (rows, 6) float64"""
    rows = theta_partial.numel() // 6
    ws = workspace(L().dspn_affine_sampler_theta_reduce_workspace_bytes(rows), theta_partial.device, "theta_reduce")
    assert theta_partial.dtype == torch.float64 and dtheta.dtype == torch.float32 and dtheta.numel() == 6
    check(L().dspn_affine_sampler_theta_reduce(ptr(theta_partial), rows, ptr(dtheta), int(accumulate), ptr(ws), ws.numel(),
                                               stream()), "affine_sampler_theta_reduce")
    return dtheta


def affine_sampler_backward_theta(sources, theta, dy, dtheta, accumulate=False):
    N, Ho, Wo, ldo = dy.shape
    ws = workspace(L().dspn_affine_sampler_theta_workspace_bytes(N, Ho, Wo), dy.device, "theta")
    assert all(t.dtype == dy.dtype for t in sources.tensors) and dtheta.dtype == torch.float32
    check(_f("dspn_affine_sampler_backward_theta", dy)(sources.x, sources.Hin, sources.Win, sources.C, sources.coff,
                                                     sources.n, ptr(theta), ptr(dy), N, Ho, Wo, ldo, ptr(dtheta),
                                                     int(accumulate), ptr(ws), ws.numel(), stream()),
          "affine_sampler_backward_theta")
    return dtheta


def seg_counts(scores, label, C):
    """scores (..., ld >= C) class scores per pixel, label (...) -> int64 tensor [3C+1]: per class intersection,
    predicted count, label count, then the number of pixels with pred == label"""
    ld = scores.shape[-1]
    rows = _rows(scores)
    assert label.numel() == rows
    out = torch.zeros(3 * C + 1, dtype=torch.int64, device=scores.device)
    check(L().dspn_seg_counts_f32(ptr(scores), ptr(label), rows, C, ld, ptr(out), stream()), "seg_counts")
    return out


def seg_upsample_argmax(prob, C, Ho, Wo):
    """prob (N, Hin, Win, ld >= C) NHWC class probabilities -> uint8 (N, Ho, Wo): bilinear sampling on the identity
    grid fused with the argmax over classes (multi_eval.py:28-34)"""
    assert prob.dim() == 4 and prob.is_contiguous() and prob.dtype == torch.float32
    N, Hin, Win, ld = prob.shape
    out = torch.empty(N, Ho, Wo, dtype=torch.uint8, device=prob.device)
    check(L().dspn_seg_upsample_argmax_f32(ptr(prob), ptr(out), N, Hin, Win, C, ld, Ho, Wo, stream()),
          "seg_upsample_argmax")
    return out


# ------------------------------------------------------------------ losses / optimizer
def softmax_output(logits, label, C, ignore_label, grad_scale=1.0, valid_count=None, prob=None, grad=None,
                   want_grad=True):
    ld = logits.shape[-1]
    rows = _rows(logits)
    prob = torch.empty_like(logits, dtype=torch.float32) if prob is None else prob
    if want_grad and grad is None:
        grad = torch.empty_like(logits)
    assert prob.dtype == torch.float32 and (grad is None or not want_grad or grad.dtype == logits.dtype)
    check(_f("dspn_softmax_output", logits)(ptr(logits), ptr(label), ptr(prob), ptr(grad) if want_grad else 0, rows,
                                      C, ld, float(ignore_label), float(grad_scale), ptr(valid_count),
                                      stream()), "softmax_output")
    return prob, grad


def count(a, mode, ref, out=None):
    """mode 'ne': #(a != ref); mode 'gt': #(a > ref); result is a 1-element device tensor (not clamped)"""
    out = empty(1, device=a.device) if out is None else out
    check(L().dspn_count_f32(ptr(a), a.numel(), 0 if mode == "ne" else 1, float(ref), ptr(out), stream()),
          "count")
    return out


def smooth_l1_forward(pred, target, mask, out=None):
    out = torch.empty_like(pred) if out is None else out
    check(L().dspn_smooth_l1_forward_f32(ptr(pred), ptr(target), ptr(mask), ptr(out), pred.numel(), stream()),
          "smooth_l1_forward")
    return out


def smooth_l1_backward(pred, target, mask, valid_count, grad_scale=1.0, out=None):
    out = torch.empty_like(pred) if out is None else out
    check(L().dspn_smooth_l1_backward_f32(ptr(pred), ptr(target), ptr(mask), ptr(out), pred.numel(),
                                          float(grad_scale), ptr(valid_count), stream()), "smooth_l1_backward")
    return out


def cross_entropy_sum(prob, label, C, ignore_label, eps=1e-8, out=None):
    out = empty(2, device=prob.device) if out is None else out
    check(L().dspn_cross_entropy_sum_f32(ptr(prob), ptr(label), _rows(prob), C, prob.shape[-1],
                                         float(ignore_label), float(eps), ptr(out), stream()),
          "cross_entropy_sum")
    return out


def sum_all(a, out=None):
    out = empty(1, device=a.device) if out is None else out
    check(L().dspn_sum_f32(ptr(a), a.numel(), ptr(out), stream()), "sum")
    return out


def sgd_momentum(w, grad, mom, lr, momentum, wd, rescale):
    check(L().dspn_sgd_momentum_f32(ptr(w), ptr(grad), ptr(mom), w.numel(), lr, momentum, wd, rescale,
                                    stream()), "sgd_momentum")
