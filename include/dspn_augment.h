/*
 * dspn_augment.h -- C ABI of the per-batch image pipeline of liangfu/dspnet's MultiTaskRecordIter
 * (dataset/iterator.py:301-603): what `_get_augmented` / `_get_resized` (:412-548)
 * and the tail of `_get_batch` (:568-576) do to the decoded image and its segmentation map, fused into one launch per
 * batch on the device.  Decoding (JPEG / PNG) and the box arithmetic stay on the host.
 *
 * Per sample the reference runs (OpenCV, third-party, version unpinned):
 *     img = cv2.warpAffine(img, M, (W, H), INTER_LINEAR,  borderValue = 128 (augmented) | 0 (resized))
 *     seg = cv2.warpAffine(seg, M, (W, H), INTER_NEAREST, borderValue = 255 (augmented) | 0 (resized))
 *     if flip: img = cv2.flip(img, 1); seg = cv2.flip(seg, 1)
 *     data[c] = img[:, :, 2 - c] - mean[c]                       (BGR -> RGB planes, float64 subtract, float32 store)
 *     seg = cv2.resize(seg, (W/4, H/4), INTER_NEAREST); seg = LUT(seg).astype(uint8)   -> seg_out_label (float32)
 * The kernels restate OpenCV's integer arithmetic exactly (imgwarp.cpp: warpAffine / remap):
 *     inverse map in double on the host (warpAffine inverts M unless WARP_INVERSE_MAP);
 *     adelta[x] = cvRound(m0 * x * 1024), bdelta[x] = cvRound(m3 * x * 1024),
 *     X0 = cvRound((m1 * y + m2) * 1024) + rd, Y0 = cvRound((m4 * y + m5) * 1024) + rd,  cvRound = round-half-even;
 *     nearest : rd = 512, sx = (X0 + adelta[x]) >> 10, sy likewise; outside the image -> border value;
 *     bilinear: rd = 16,  X = (X0 + adelta[x]) >> 5, sx = X >> 5, fx = X & 31 (same for y); integer weights
 *               32 * (32 - fy | fy) * (32 - fx | fx) (sum 32768); each of the 4 taps outside the image reads the
 *               border value; result = (sum w * tap + 16384) >> 15.
 *     INTER_NEAREST resize by exactly 1/4 reads source pixel (4y, 4x).
 * PARITY STATUS: unpinned (no OpenCV in this image to produce vectors).
 *
 * Conventions as in dspn_multibox.h: device pointers, caller-owned buffers, explicit stream, status return +
 * dspn_last_error().
 */
#ifndef DSPN_AUGMENT_H_
#define DSPN_AUGMENT_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one decoded sample inside the batch's byte pools */
typedef struct dspn_warp_sample {
  long long img_offset;     /* byte offset of the (src_h, src_w, 3) uint8 image in `images` */
  long long seg_offset;     /* byte offset of the (src_h, src_w) uint8 label map in `segs`; < 0: no map (labels stay 0) */
  int src_h, src_w;
  int flip;                 /* != 0: horizontal flip after the warp */
  int img_border;           /* border value of the image warp (all channels), 0..255 */
  int seg_border;           /* border value of the label warp */
  int reserved;
  double minv[6];           /* INVERSE affine map (destination -> source), row major 2 x 3, as OpenCV derives it */
} dspn_warp_sample;

/* images / segs: device byte pools; samples: B descriptors in DEVICE memory; channel_map[c] = source channel that
 * becomes output plane c (BGR source, RGB planes: {2, 1, 0}); mean[c] is subtracted from plane c in double;
 * lut: 256 device bytes applied to the label after the quarter-size resize (NULL = identity).
 * data_out: (B, 3, H, W) float32; seg_out: (B, H/4, W/4) float32 (NULL to skip).  H, W multiples of 4. */
int dspn_augment_batch_u8(const unsigned char *images, const unsigned char *segs, const dspn_warp_sample *samples,
                          int B, int H, int W, const int channel_map[3], const double mean[3],
                          const unsigned char *lut, float *data_out, float *seg_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif  /* DSPN_AUGMENT_H_ */
