/*
 * dspn_multibox.h -- C ABI of the MI355X (gfx950) SSD multibox operators.
 *
 * Drop-in boundary for the three custom operators liangfu/dspnet compiles into
 * MXNet.  Each entry point replaces the `Forward` of one
 * operator class; the reference interface it stands in for is cited per
 * function.  The caller (the framework) owns every buffer including the
 * workspace, exactly as MXNet owns the TBlobs and the kTempSpace resource
 * (operator/multibox_target-inl.h:118-120, :258-261).  Kernels never
 * allocate, never synchronise, and launch on the stream they are given.
 *
 * All pointers named *_dev are DEVICE pointers to dense row-major float32.
 * Small parameter arrays (sizes, ratios, variances) are HOST pointers, read
 * during the call (they are operator attributes in the reference, not
 * tensors).  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *
 * Return: 0 on success, negative dspn_status on failure; the message is kept in
 * a thread-local string readable through dspn_last_error().
 *
 * Gradients: all three operators are constants in the backward pass (the
 * reference writes zeros: multibox_prior-inl.h:131-143, multibox_target-inl.h:
 * 173-185, multibox_detection-inl.h:109-125); there are no backward entries.
 */
#ifndef DSPN_MULTIBOX_H_
#define DSPN_MULTIBOX_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum dspn_status {
  DSPN_OK = 0,
  DSPN_ERR_ARG = -1,        /* a reference CHECK on shapes/attributes would fail */
  DSPN_ERR_WORKSPACE = -2,  /* workspace too small */
  DSPN_ERR_LAUNCH = -3      /* hipGetLastError() after a launch */
};

/* message of the last failing call on this thread ("" if none) */
const char *dspn_last_error(void);
/* library / ABI version, bumped when a signature changes */
int dspn_abi_version(void);
/* Optional kernel timing with HIP events on the launch stream (off by default).  Families:
 * 0 = implicit-GEMM conv forward / data-gradient kernel, 1 = conv weight-gradient (+ slab reduce).
 * dspn_profile_collect synchronises the recorded events, returns their summed duration and count
 * for one family and forgets them. */
int dspn_profile_enable(int on);
int dspn_profile_collect(int family, double *total_ms, long long *launches);
/* Launch setting for data-parallel runs (round 4; not compute state -- every result is bit-identical under every value):
 * the persistent convolution kernels size their grid to fill the chip, so the kernels of another queue (RCCL's all-reduce
 * of a gradient bucket) only start between two convolution launches.  With cus > 0 those grids leave `cus` CUs' worth of
 * workgroup slots free (0 .. 128; 0 = default).  bench.py --gpus N switches it on after warm-up only if the measured
 * exposed all-reduce time says the collectives are not being hidden. */
int dspn_conv_set_reserved_cus(int cus);
/* Launch setting (round 5; not compute state): which tile family the two-piece convolutions whose operands are both piece
 * planes run on.  0 = automatic (by layer shape), 1 = never the wide family (conv_nt_kernel's 128 x 128 tiles, the round-4
 * schedule), 2 / 3 / 4 = always 256 x 128 / 128 x 256 / 128 x 128-on-four-waves where legal (tests and experiments).  The K
 * order and the per-output accumulation order are the same on every tile. */
int dspn_conv_set_wide_tiles(int mode);
/* Launch setting (round 6; not compute state): the round-6 loops of the 128-row members of the wide convolution family on layers
 * with a dense output of whole 128-row tiles.  1 (default; the environment variable DSPN_XT sets the process's initial value):
 * the plane-fed kernels (128 x 128 and 256 x 64 on four waves, 128 x 256 on eight) run their TILE-SPANNING loop -- the next tile's first
 * operand images are requested while the current tile is written out -- with the epilogue straight from the accumulators, and
 * the float-operand kernels run that direct epilogue on their round-5 loop.  0 = the round-5 kernels as they were; 2 = also the
 * float-operand kernel's tile-spanning loop (measured slower inside the training step: experiments).  Stored tensors, per-tile
 * extremes and magnitude blocks are the same bits under every value; the per-tile BatchNorm tables agree within fp32 rounding
 * (another summation order). */
int dspn_conv_set_tile_spanning(int on);
/* Launch setting (round 6; not compute state): 1 (default) = the data gradient of the affine sampler
 * (dspn_affine_sampler_backward_data_theta_*) takes the geometry of a source position once for the whole batch (one workgroup
 * per position, the matching target rows of four matches requested together); 0 = one workgroup per source pixel of every
 * image (the round-4 kernel).  The same bits either way (tests). */
int dspn_affine_sampler_set_batched(int on);

/* Replaces MultiBoxPriorOp::Forward (operator/multibox_prior-inl.h:97-129) +
 * MultiBoxPriorForward (operator/multibox_prior.cc:30-71; GPU twin
 * multibox_prior.cu:39-101).  out_dev: (in_height*in_width*(num_sizes+num_ratios-1), 4).
 * step_* <= 0 -> 1/in_height, 1/in_width (inl.h:119-123).  */
int dspn_multibox_prior_f32(const float *sizes, int num_sizes,
                            const float *ratios, int num_ratios,
                            int in_height, int in_width,
                            float step_y, float step_x,
                            float offset_y, float offset_x, int clip,
                            float *out_dev, void *stream);

/* Workspace for dspn_multibox_target_f32 (bytes). */
size_t dspn_multibox_target_workspace_bytes(int batch, int num_anchors, int num_labels);

/* Replaces MultiBoxTargetOp::Forward (operator/multibox_target-inl.h:89-171) +
 * MultiBoxTargetForward (operator/multibox_target.cc:73-284; GPU twin
 * multibox_target.cu:343-419).
 *   anchors_dev   (num_anchors, 4)          [xmin,ymin,xmax,ymax]
 *   labels_dev    (batch, num_labels, label_width) rows [cls,xmin,ymin,xmax,ymax,dist], -1 padded
 *   cls_preds_dev (batch, num_classes, num_anchors)   (class 0 = background)
 *   loc_target_dev, loc_mask_dev (batch, num_anchors*5); cls_target_dev (batch, num_anchors)
 * Outputs are fully overwritten.  CPU semantics are followed where the
 * reference's CPU and CUDA paths disagree (minimum_negative_samples is accepted
 * and ignored, as in multibox_target.cc:186-189).
 * Data-dependent reference aborts (multibox_target.cc:98-101, :236) cannot
 * abort a stream: they set a per-sample code in the first `batch` ints of the
 * workspace (0 ok, 2 = padded label row not all -1, 3 = fewer mining candidates
 * than requested negatives); see dspn_multibox_target_errors(). */
int dspn_multibox_target_f32(const float *anchors_dev, const float *labels_dev,
                             const float *cls_preds_dev,
                             int batch, int num_anchors, int num_labels,
                             int label_width, int num_classes,
                             float overlap_threshold, float ignore_label,
                             float negative_mining_ratio,
                             float negative_mining_thresh,
                             int minimum_negative_samples,
                             const float variances[4],
                             float *loc_target_dev, float *loc_mask_dev,
                             float *cls_target_dev,
                             void *workspace_dev, size_t workspace_bytes,
                             void *stream);

/* Copies the per-sample error codes of the last target call that used this
 * workspace to host_codes[batch] (synchronises `stream`).  Returns the first
 * non-zero code negated, or 0. */
int dspn_multibox_target_errors(const void *workspace_dev, int batch,
                                int *host_codes, void *stream);

/* Workspace for dspn_multibox_detection_f32 (bytes). */
size_t dspn_multibox_detection_workspace_bytes(int batch, int num_anchors);

/* Replaces MultiBoxDetectionOp::Forward (operator/multibox_detection-inl.h:81-107)
 * + MultiBoxDetectionForward (operator/multibox_detection.cc:54-169; GPU twin
 * multibox_detection.cu:53-236).
 *   cls_prob_dev (batch, num_classes, num_anchors); loc_pred_dev (batch, num_anchors*5)
 *   anchors_dev (num_anchors, 4); out_dev (batch, num_anchors, 7) =
 *   [id, score, xmin, ymin, xmax, ymax, dist], id = -1 for empty/suppressed rows.
 * CPU semantics: rows are compacted in anchor order, stable-sorted by score,
 * only the first nms_topk sorted rows are written back, NMS runs over all valid
 * rows (multibox_detection.cc:143-167). */
int dspn_multibox_detection_f32(const float *cls_prob_dev, const float *loc_pred_dev,
                                const float *anchors_dev,
                                int batch, int num_anchors, int num_classes,
                                float threshold, int clip,
                                const float variances[4],
                                float nms_threshold, int force_suppress,
                                int nms_topk,
                                float *out_dev,
                                void *workspace_dev, size_t workspace_bytes,
                                void *stream);

#ifdef __cplusplus
}
#endif
#endif  /* DSPN_MULTIBOX_H_ */
