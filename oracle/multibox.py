"""numpy front end of oracle/multibox_oracle.c (ctypes).  TEST INFRASTRUCTURE ONLY.

Builds oracle/_build/libdspn_oracle.so with `make` on first use (gcc only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdspn_oracle.so")
_lib = None
_f32p = ctypes.POINTER(ctypes.c_float)


def build():
    src = os.path.join(_HERE, "multibox_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


_lib_expd = None


def lib_exp_double():
    """the oracle built with -DDSPN_ORACLE_EXP_DOUBLE (the other reading of multibox_detection.cc:113)"""
    global _lib_expd
    if _lib_expd is None:
        so = os.path.join(_HERE, "_build", "libdspn_oracle_expd.so")
        src = os.path.join(_HERE, "multibox_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib_expd = ctypes.CDLL(so)
    return _lib_expd


def _p(a):
    return a.ctypes.data_as(_f32p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def multibox_prior(in_height, in_width, sizes, ratios, clip=False, steps=(-1.0, -1.0),
                   offsets=(0.5, 0.5)):
    sizes, ratios = _f32(sizes), _f32(ratios)
    n = in_height * in_width * (len(sizes) + len(ratios) - 1)
    out = np.empty((1, n, 4), np.float32)
    rc = lib().dspn_oracle_multibox_prior(_p(sizes), len(sizes), _p(ratios), len(ratios),
                                          ctypes.c_int(in_height), ctypes.c_int(in_width),
                                          ctypes.c_float(steps[0]), ctypes.c_float(steps[1]),
                                          ctypes.c_float(offsets[0]), ctypes.c_float(offsets[1]),
                                          int(bool(clip)), _p(out))
    if rc != 0:
        raise ValueError(f"oracle multibox_prior rc={rc}")
    return out


def multibox_target(anchor, label, cls_pred, overlap_threshold=0.5, ignore_label=-1.0,
                    negative_mining_ratio=-1.0, negative_mining_thresh=0.5,
                    minimum_negative_samples=0, variances=(0.1, 0.1, 0.2, 0.2),
                    return_code=False):
    anchor, label, cls_pred = _f32(anchor), _f32(label), _f32(cls_pred)
    B, L, lw = label.shape
    A = anchor.shape[-2]
    C = cls_pred.shape[1]
    var = _f32(variances)
    loc_t = np.empty((B, A * 5), np.float32)
    loc_m = np.empty((B, A * 5), np.float32)
    cls_t = np.empty((B, A), np.float32)
    rc = lib().dspn_oracle_multibox_target(
        _p(anchor), _p(label), _p(cls_pred), B, A, L, lw, C,
        ctypes.c_float(overlap_threshold), ctypes.c_float(ignore_label),
        ctypes.c_float(negative_mining_ratio), ctypes.c_float(negative_mining_thresh),
        int(minimum_negative_samples), _p(var), _p(loc_t), _p(loc_m), _p(cls_t))
    if return_code:
        return [loc_t, loc_m, cls_t], rc
    if rc != 0:
        raise ValueError(f"oracle multibox_target rc={rc}")
    return [loc_t, loc_m, cls_t]


def multibox_detection(cls_prob, loc_pred, anchor, clip=True, threshold=0.01, background_id=0,
                       nms_threshold=0.5, force_suppress=False, variances=(0.1, 0.1, 0.2, 0.2),
                       nms_topk=-1, exp_double=False):
    cls_prob, loc_pred, anchor = _f32(cls_prob), _f32(loc_pred), _f32(anchor)
    B, C, A = cls_prob.shape
    var = _f32(variances)
    out = np.empty((B, A, 7), np.float32)
    rc = (lib_exp_double() if exp_double else lib()).dspn_oracle_multibox_detection(
        _p(cls_prob), _p(loc_pred), _p(anchor), B, A, C, ctypes.c_float(threshold),
        int(bool(clip)), _p(var), ctypes.c_float(nms_threshold), int(bool(force_suppress)),
        int(nms_topk), _p(out))
    if rc != 0:
        raise ValueError(f"oracle multibox_detection rc={rc}")
    return out
