/*
 * dspn_nms.h -- C ABI of the pixel-coordinate (Fast R-CNN, "+1" convention) non-maximum suppression of
 * liangfu/dspnet: detect/nms.py:24-58 (`nms`, the one the video demo calls at
 * detect/multitask_detector.py:450), cython/cpu_nms.pyx:17-68 (`cpu_nms`) and cython/nms_kernel.cu:24-144 +
 * cython/gpu_nms.pyx (`gpu_nms`), and of the box-overlap matrix of cython/bbox.pyx:15-55 (`bbox_overlaps_cython`, at the
 * end of this file).  SURVEY.md section 8f rank 4.
 *
 * All three compute, for boxes [x1, y1, x2, y2, score] taken in descending score order,
 *     area = (x2 - x1 + 1) * (y2 - y1 + 1),  inter = max(0, xx2 - xx1 + 1) * max(0, yy2 - yy1 + 1),
 *     ovr  = inter / (area_i + area_j - inter)
 * in float32 and keep a box unless an already kept one overlaps it; they differ in the comparison:
 *     nms() keeps `ovr <= thresh` (detect/nms.py:55), i.e. drops when NOT(ovr <= thresh):                suppress_ge = 0
 *     cpu_nms suppresses `ovr >= thresh` (cpu_nms.pyx:65):                                                suppress_ge = 1
 *     nms_kernel suppresses `ovr > thresh` (nms_kernel.cu:68):                                            suppress_ge = 2
 *   modes 0 and 2 differ only for a NaN overlap (two zero-area boxes, 0/0): numpy's comparison drops the pair, the
 *   CUDA kernel's keeps it (pinned by tests/golden/nms_pixel.npz, case degenerate_zero_union).
 * Equal scores: the reference sorts with numpy's unstable argsort()[::-1]; here ties go to the HIGHER index first
 * (the reverse of a stable ascending sort), which is what numpy returns for short arrays.
 *
 * Conventions as in dspn_multibox.h: device pointers, caller-owned buffers and workspace, explicit stream, status
 * return + dspn_last_error().
 */
#ifndef DSPN_NMS_H_
#define DSPN_NMS_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* n <= 8192 boxes */
size_t dspn_nms_pixel_workspace_bytes(int n);

/* dets_dev: (n, 5) float32.  keep_dev: n int32 slots receiving the ORIGINAL indices of the kept boxes in
 * descending score order (the reference's `keep` list); num_keep_dev: one int32. */
int dspn_nms_pixel_f32(const float *dets_dev, int n, float thresh, int suppress_ge, int *keep_dev,
                       int *num_keep_dev, void *workspace, size_t workspace_bytes, void *stream);

/* cython/bbox.pyx:15-55 (`bbox_overlaps_cython`): overlaps (N, K) float64, row-major, of boxes (N, 4) against
 * query_boxes (K, 4) [x1, y1, x2, y2] float64 in the same "+1" convention:
 *     iw = min(x2, qx2) - max(x1, qx1) + 1;  ih likewise;  both > 0:  iw * ih / (area + query_area - iw * ih),  else 0
 * in float64, the reference's operation order (bit-identical to it: tests/golden/bbox_overlaps.npz).  N or K == 0: no-op. */
int dspn_bbox_overlaps_f64(const double *boxes_dev, int N, const double *query_dev, int K, double *overlaps_dev,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif  /* DSPN_NMS_H_ */
