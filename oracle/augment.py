"""numpy restatement of the per-sample work of MultiTaskRecordIter (dataset/iterator.py:412-576): the OpenCV calls
(cv2.warpAffine linear / nearest with a constant border, cv2.flip, cv2.resize nearest, cv2.LUT, cv2.transform) and the
box bookkeeping of `_get_augmented` / `_get_resized`, the latter in the reference's own statement order.
TEST INFRASTRUCTURE ONLY.

PARITY STATUS: "parity unpinned".  OpenCV is a third-party dependency of the reference (`import cv2`, version
unpinned) and is not installed here; the integer arithmetic of warpAffine / remap is restated from OpenCV's published
implementation (modules/imgproc/src/imgwarp.cpp: AB_BITS = 10, INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15) and
checked in tests/test_record_iter.py only against properties that do not depend on that reading (identity and
integer-shift warps reproduce the source; agreement with a float bilinear interpolation within 1 grey level)."""
import math

import numpy as np


def invert_affine(M):
    """warpAffine without WARP_INVERSE_MAP: the 2x3 matrix is inverted in double before use"""
    M = np.array(M, np.float64).reshape(2, 3).copy()
    m = M.reshape(-1)
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2] = b1; m[5] = b2
    return m.copy()


def _cv_round(v):
    return np.rint(v).astype(np.int64)      # cvRound: nearest, ties to even


def warp_affine(src, M, dsize, linear, border):
    """cv2.warpAffine(src, M, dsize=(W, H), flags=INTER_LINEAR|INTER_NEAREST, borderMode=CONSTANT, borderValue=border)
    for uint8 src (h, w) or (h, w, c)"""
    W, H = dsize
    m = invert_affine(M)
    h, w = src.shape[:2]
    s3 = src.reshape(h, w, -1).astype(np.int64)
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    adelta = _cv_round(m[0] * x * 1024.0)
    bdelta = _cv_round(m[3] * x * 1024.0)
    rd = 16 if linear else 512
    X0 = _cv_round((m[1] * y + m[2]) * 1024.0) + rd
    Y0 = _cv_round((m[4] * y + m[5]) * 1024.0) + rd
    XX = X0[:, None] + adelta[None, :]
    YY = Y0[:, None] + bdelta[None, :]

    def tap(sy, sx):
        ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
        v = s3[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)]
        return np.where(ok[..., None], v, border)

    if not linear:
        sx = np.clip(XX >> 10, -32768, 32767)
        sy = np.clip(YY >> 10, -32768, 32767)
        out = tap(sy, sx)
    else:
        X = XX >> 5
        Y = YY >> 5
        sx = np.clip(X >> 5, -32768, 32767)
        sy = np.clip(Y >> 5, -32768, 32767)
        fx = (X & 31)[..., None]
        fy = (Y & 31)[..., None]
        acc = tap(sy, sx) * (32 * (32 - fy) * (32 - fx)) + tap(sy, sx + 1) * (32 * (32 - fy) * fx) \
            + tap(sy + 1, sx) * (32 * fy * (32 - fx)) + tap(sy + 1, sx + 1) * (32 * fy * fx)
        out = (acc + (1 << 14)) >> 15
    return out.astype(np.uint8).reshape((H, W) + src.shape[2:])


def transform_points(pts, M):
    """cv2.transform for (1, n, 2) points and a 2x3 matrix (double)"""
    M = np.asarray(M, np.float64)
    p = np.asarray(pts, np.float64)
    return p @ M[:, :2].T + M[:, 2]


def seg_lut():
    """dataset/iterator.py:361-366 with dataset/cs_labels.py:63-98: identity for the ids that carry a trainId >= 0
    (0..34), 255 elsewhere"""
    lut = np.ones(256) * 255
    lut[:35] = np.arange(35)
    return lut


def _boxes_to_top(hdr_reshaped, xmax):
    """:466-469 / :541-544 including the squeeze: exactly one surviving box comes back 1-d and is broadcast
    into the first six rows"""
    idx_valid = np.where(xmax > -.5)
    reshaped_top = np.squeeze(hdr_reshaped[np.asarray(idx_valid), :])
    hdr_reshaped.fill(-1)
    hdr_reshaped[:reshaped_top.shape[0], :] = reshaped_top


def get_resized(img, hdr, seg, data_shape):
    """dataset/iterator.py:436-473 (`_get_resized`); hdr is modified in place like the reference's view"""
    hh, ww, ch = img.shape
    theta, sx, sy, tx, ty = 0., 1. * (data_shape[2] / float(ww)), 1. * (data_shape[1] / float(hh)), 0, 0
    M = np.array([[sx * math.cos(theta), -sy * math.sin(theta), tx], [sx * math.sin(theta), sy * math.cos(theta), ty]])
    img = warp_affine(img, M, (data_shape[2], data_shape[1]), True, 0)
    if seg is not None:
        seg = warp_affine(seg, M, (data_shape[2], data_shape[1]), False, 0)
    hdr_reshaped = hdr[3:].reshape((-1, 6))
    cls = hdr_reshaped[:, 0]
    idx = np.where(cls >= 0)
    if idx[0].shape[0] < 1:
        return img, hdr, seg
    xmin, ymin, xmax, ymax = hdr_reshaped[:, 1], hdr_reshaped[:, 2], hdr_reshaped[:, 3], hdr_reshaped[:, 4]
    areas = (xmax - xmin) * data_shape[2] * (ymax - ymin) * data_shape[1]
    hdr_reshaped[np.where(areas < 100)[0], :] = -1
    _boxes_to_top(hdr_reshaped, xmax)
    return img, hdr, seg


def get_augmented(img, hdr, seg, data_shape, aug_args):
    """dataset/iterator.py:475-548 (`_get_augmented`); aug_args = (flip, theta, sx, sy, tx, ty)"""
    hh, ww, ch = img.shape
    flip, theta, sx, sy, tx, ty = tuple(aug_args)
    sx2, sy2 = sx * (data_shape[2] / float(ww)), sy * (data_shape[1] / float(hh))
    M = np.array([[sx2 * math.cos(theta), -sy2 * math.sin(theta), tx], [sx2 * math.sin(theta), sy2 * math.cos(theta), ty]])
    img = warp_affine(img, M, (data_shape[2], data_shape[1]), True, 128)
    seg = warp_affine(seg, M, (data_shape[2], data_shape[1]), False, 255)
    cls = hdr[3:].reshape((-1, 6))[:, 0]
    idx = np.where(cls >= 0)
    hdr_reshaped = hdr[3:].reshape((-1, 6))
    pts = hdr_reshaped[:, 1:5]
    dist = hdr_reshaped[idx[0], 5]
    n = idx[0].shape[0]
    xop = np.ones((n, 1)) * data_shape[2]
    yop = np.ones((n, 1)) * data_shape[1]
    pts_new = pts[idx] * np.hstack((xop, yop, xop, yop))
    if pts_new.shape[0] < 1:
        return img, hdr, seg                                   # (before the flip: an image without boxes is never flipped)
    pts_new = np.expand_dims(np.vstack((pts_new[:, :2], pts_new[:, 2:])), axis=0)
    M = np.array([[sx * math.cos(theta), -sy * math.sin(theta), tx], [sx * math.sin(theta), sy * math.cos(theta), ty]])
    pts_new = np.squeeze(transform_points(pts_new, M))
    if flip > .5:
        pts_new[:, 0] = data_shape[2] - pts_new[:, 0]
    pts_new = pts_new * np.hstack((np.ones((pts_new.shape[0], 1)) / data_shape[2],
                                   np.ones((pts_new.shape[0], 1)) / data_shape[1]))
    pts_new = np.hstack((pts_new[:n, :], pts_new[n:, :]))
    if flip > .5:
        tmp = pts_new[:, 2].copy()
        pts_new[:, 2] = pts_new[:, 0]
        pts_new[:, 0] = tmp
    xmin, ymin, xmax, ymax = pts_new[:, 0], pts_new[:, 1], pts_new[:, 2], pts_new[:, 3]
    idx_small = np.where(xmax > -.5)
    xmin[idx_small] = np.clip(xmin[idx_small], 0, 1)
    xmax[idx_small] = np.clip(xmax[idx_small], 0, 1)
    ymin[idx_small] = np.clip(ymin[idx_small], 0, 1)
    ymax[idx_small] = np.clip(ymax[idx_small], 0, 1)
    hdr_reshaped[idx[0], 1:5] = pts_new
    hdr_reshaped[idx[0], 5] = dist / math.sqrt(sx * sy)
    xmin, ymin, xmax, ymax = hdr_reshaped[:, 1], hdr_reshaped[:, 2], hdr_reshaped[:, 3], hdr_reshaped[:, 4]
    areas = (xmax - xmin) * data_shape[2] * (ymax - ymin) * data_shape[1]
    hdr_reshaped[np.where(areas < 100)[0], :] = -1
    hdr_reshaped[np.where(xmax < .01)[0], :] = -1
    hdr_reshaped[np.where(xmin > .99)[0], :] = -1
    hdr_reshaped[np.where(ymax < .01)[0], :] = -1
    hdr_reshaped[np.where(ymin > .99)[0], :] = -1
    _boxes_to_top(hdr_reshaped, xmax)
    if flip > .5:
        img = img[:, ::-1].copy()
        seg = seg[:, ::-1].copy()
    return img, hdr, seg


def finish_sample(img, seg, data_shape, mean_pixels, lut):
    """dataset/iterator.py:568-575: planes 2-c minus mean (float64 -> float32), label map /4 nearest + LUT"""
    data = np.zeros((3, data_shape[1], data_shape[2]))
    for chidx in range(3):
        data[chidx] = img[:, :, 2 - chidx] - mean_pixels[chidx]
    seg_out = None
    if seg is not None:
        q = seg[::4, ::4]                                       # cv2.resize(..., INTER_NEAREST) by exactly 1/4
        seg_out = lut[q].astype(np.uint8).astype(np.float32)
    return data.astype(np.float32), seg_out
