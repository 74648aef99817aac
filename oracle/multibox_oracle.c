/*
 * multibox_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the three SSD operators of liangfu/dspnet, used
 * ONLY as the checker in tests/, __graft_entry__.smoke() and as the
 * `cpu_baseline` leg of bench.py.  Nothing under dspnet_amd/ may import, link
 * or call it.
 *
 * PARITY STATUS: "parity unpinned".  The reference ships no tests, golden
 * vectors or fixtures for these operators, its Python
 * needs Python 2 + an un-vendored MXNet, and its C++ operator files cannot be
 * compiled here without writing stand-ins for the MXNet/mshadow/dmlc headers
 * they include (operator/multibox_target-inl.h:30-38), which this project does
 * not do.  The only reference-produced numbers available are the three probe
 * outputs recorded in SURVEY.md section 8(c); tests/test_oracle_multibox.py
 * checks this file against them.  Everything else is a line-by-line reading
 * of the reference sources cited at each function.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the reference is
 * x86-64 baseline code with no FMA contraction, and the bit patterns of the
 * IoU / softmax values decide match and NMS indices).
 *
 * All tensors are dense row-major float32, host pointers.
 * Return value: 0 ok; <0 = a reference CHECK would have aborted:
 *   -1 bad argument, -2 padded label row not all -1, -3 fewer mining
 *   candidates than requested negatives.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* MultiBoxPrior                                                       */
/* ------------------------------------------------------------------ */
/* follows operator/multibox_prior.cc:30-71 (box loop) and
 * operator/multibox_prior-inl.h:118-128 (step defaulting, optional clip) */
int dspn_oracle_multibox_prior(const float *sizes, int num_sizes,
                               const float *ratios, int num_ratios,
                               int in_height, int in_width,
                               float step_y, float step_x,
                               float off_y, float off_x, int clip,
                               float *out /* (H*W*(ns+nr-1), 4) */) {
  if (num_sizes <= 0 || num_ratios <= 0 || in_height <= 0 || in_width <= 0)
    return -1;
  if (step_y * step_x < 0) return -1;      /* -inl.h:118 */
  if (step_y <= 0 || step_x <= 0) {        /* -inl.h:119-123 */
    step_y = 1.f / in_height;
    step_x = 1.f / in_width;
  }
  float *o = out;
  for (int r = 0; r < in_height; ++r) {
    float cy = (r + off_y) * step_y;
    for (int c = 0; c < in_width; ++c) {
      float cx = (c + off_x) * step_x;
      for (int i = 0; i < num_sizes; ++i) {          /* ratio 1, every size */
        float size = sizes[i];
        float w = size * in_height / in_width / 2;
        float h = size / 2;
        o[0] = cx - w; o[1] = cy - h; o[2] = cx + w; o[3] = cy + h;
        o += 4;
      }
      float size0 = sizes[0];
      for (int j = 1; j < num_ratios; ++j) {         /* size[0], other ratios */
        float ratio = sqrtf(ratios[j]);
        float w = size0 * in_height / in_width * ratio / 2;
        float h = size0 / ratio / 2;
        o[0] = cx - w; o[1] = cy - h; o[2] = cx + w; o[3] = cy + h;
        o += 4;
      }
    }
  }
  if (clip) {                                        /* -inl.h:126-128 */
    long n = (long)(o - out);
    for (long i = 0; i < n; ++i) {
      float v = out[i];
      out[i] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* MultiBoxTarget                                                      */
/* ------------------------------------------------------------------ */
static inline float fmax_(float a, float b) { return a > b ? a : b; }
static inline float fmin_(float a, float b) { return a < b ? a : b; }

/* operator/multibox_target-inl.h:137-161: intersection / union with
 * safe_divide (:44-50).  box = (l,t,r,b). */
static float target_iou(const float *a, const float *g) {
  float iw = fmax_(0.f, fmin_(a[2], g[2]) - fmax_(a[0], g[0]));
  float ih = fmax_(0.f, fmin_(a[3], g[3]) - fmax_(a[1], g[1]));
  float inter = iw * ih;
  float uni = (a[2] - a[0]) * (a[3] - a[1]) + (g[2] - g[0]) * (g[3] - g[1]) - inter;
  if (uni == 0.f) return 0.f;
  return inter / uni;
}

/* operator/multibox_target.cc:31-56.  gt points at [xmin,ymin,xmax,ymax,dist].
 * The `* 0.5` and `/ 0.1` are double operations in the reference. */
static void encode_loc(const float *anchor, const float *gt, float *dst,
                       float vx, float vy, float vw, float vh) {
  float al = anchor[0], at = anchor[1], ar = anchor[2], ab = anchor[3];
  float aw = ar - al, ah = ab - at;
  float ax = (float)((al + ar) * 0.5);
  float ay = (float)((at + ab) * 0.5);
  float gl = gt[0], gtp = gt[1], gr = gt[2], gb = gt[3], gz = gt[4];
  float gw = gr - gl, gh = gb - gtp;
  float gx = (float)((gl + gr) * 0.5);
  float gy = (float)((gtp + gb) * 0.5);
  dst[0] = (gx - ax) / aw / vx;
  dst[1] = (gy - ay) / ah / vy;
  dst[2] = logf(gw / aw) / vw;
  dst[3] = logf(gh / ah) / vh;
  dst[4] = (float)((double)gz / 0.1);
}

typedef struct { float value; int index; } sort_elem;
/* std::stable_sort with operator< == (value > other.value) on elements pushed
 * in ascending index order  ==  total order (value desc, index asc). */
static int cmp_desc(const void *pa, const void *pb) {
  const sort_elem *a = (const sort_elem *)pa, *b = (const sort_elem *)pb;
  if (a->value > b->value) return -1;
  if (a->value < b->value) return 1;
  return (a->index > b->index) - (a->index < b->index);
}

/* operator/multibox_target-inl.h:89-171 (output init + IoU) and
 * operator/multibox_target.cc:73-284 (matching, mining, target write). */
int dspn_oracle_multibox_target(const float *anchors /* (A,4) */,
                                const float *labels  /* (B,L,label_w) */,
                                const float *cls_preds /* (B,Cp1,A) */,
                                int B, int A, int L, int label_w, int Cp1,
                                float overlap_threshold, float ignore_label,
                                float negative_mining_ratio,
                                float negative_mining_thresh,
                                int minimum_negative_samples /* unused on CPU, .cc:186-189 */,
                                const float variances[4],
                                float *loc_target /* (B,A*5) */,
                                float *loc_mask   /* (B,A*5) */,
                                float *cls_target /* (B,A)   */) {
  (void)minimum_negative_samples;
  if (B <= 0 || A <= 0 || L <= 0 || label_w < 6 || Cp1 <= 0) return -1;
  /* .cc:185 CHECK_GT(negative_mining_thresh, 0), inside `if (negative_mining_ratio > 0)`: the reference aborts there for the
   * first sample that has a valid ground truth.  Rejected as an argument here -- as the HIP entry does (multibox.hip) -- so
   * the checker is never laxer than the product; a batch WITHOUT any ground truth is the one input the reference lets pass. */
  if (negative_mining_ratio > 0 && !(negative_mining_thresh > 0)) return -1;
  int rc = 0;
  /* -inl.h:121-123 */
  memset(loc_target, 0, sizeof(float) * (size_t)B * A * 5);
  memset(loc_mask, 0, sizeof(float) * (size_t)B * A * 5);
  for (long i = 0; i < (long)B * A; ++i) cls_target[i] = ignore_label;

  float *overlaps = (float *)malloc(sizeof(float) * (size_t)A * L);
  float *match_iou = (float *)malloc(sizeof(float) * A);
  int *match_gt = (int *)malloc(sizeof(int) * A);
  signed char *aflag = (signed char *)malloc(A);
  char *gflag = (char *)malloc(L);
  sort_elem *cand = (sort_elem *)malloc(sizeof(sort_elem) * A);

  for (int nb = 0; nb < B; ++nb) {
    const float *lab = labels + (size_t)nb * L * label_w;
    const float *pred = cls_preds + (size_t)nb * Cp1 * A;
    /* .cc:95-105: valid rows = rows before the first class == -1 */
    int G = 0;
    for (int i = 0; i < L; ++i) {
      const float *row = lab + i * label_w;
      if (row[0] == -1.0f) {
        if (row[1] != -1.0f || row[2] != -1.0f || row[3] != -1.0f || row[4] != -1.0f)
          rc = -2;
        break;
      }
      ++G;
    }
    if (G == 0) continue;
    for (int j = 0; j < A; ++j)
      for (int k = 0; k < G; ++k)
        overlaps[(size_t)j * L + k] = target_iou(anchors + 4 * j, lab + k * label_w + 1);
    for (int j = 0; j < A; ++j) { match_iou[j] = -1.0f; match_gt[j] = -1; aflag[j] = -1; }
    memset(gflag, 0, G);
    int num_positive = 0;

    /* .cc:113-149 bipartite stage */
    for (;;) {
      int unmatched = 0;
      for (int k = 0; k < G; ++k) unmatched |= !gflag[k];
      if (!unmatched) break;
      int best_anchor = -1, best_gt = -1;
      float max_overlap = 1e-6;
      for (int j = 0; j < A; ++j) {
        if (aflag[j] == 1) continue;
        const float *row = overlaps + (size_t)j * L;
        for (int k = 0; k < G; ++k) {
          if (gflag[k]) continue;
          float iou = row[k];
          if (iou > max_overlap) { best_anchor = j; best_gt = k; max_overlap = iou; }
        }
      }
      if (best_anchor == -1) break;
      match_iou[best_anchor] = max_overlap;
      match_gt[best_anchor] = best_gt;
      num_positive += 1;
      gflag[best_gt] = 1;
      aflag[best_anchor] = 1;
    }

    /* .cc:151-180 threshold stage */
    if (overlap_threshold > 0) {
      for (int j = 0; j < A; ++j) {
        if (aflag[j] == 1) continue;
        const float *row = overlaps + (size_t)j * L;
        int best_gt = -1; float max_iou = -1.0f;
        for (int k = 0; k < G; ++k) {
          float iou = row[k];
          if (iou > max_iou) { best_gt = k; max_iou = iou; }
        }
        if (best_gt != -1) {
          match_iou[j] = max_iou; match_gt[j] = best_gt;
          if (max_iou > overlap_threshold) { num_positive += 1; gflag[best_gt] = 1; aflag[j] = 1; }
        }
      }
    }

    /* .cc:182-249 negatives */
    if (negative_mining_ratio > 0) {
      int num_negative = (int)(num_positive * negative_mining_ratio);
      if (num_negative > A - num_positive) num_negative = A - num_positive;
      if (num_negative > 0) {
        int nc = 0;
        for (int j = 0; j < A; ++j) {
          if (aflag[j] == 1) continue;
          if (match_iou[j] < 0) {            /* "not yet calculated" (.cc:198-214) */
            const float *row = overlaps + (size_t)j * L;
            int best_gt = -1; float max_iou = -1.0f;
            for (int k = 0; k < G; ++k) {
              float iou = row[k];
              if (iou > max_iou) { best_gt = k; max_iou = iou; }
            }
            if (best_gt != -1) { match_iou[j] = max_iou; match_gt[j] = best_gt; }
          }
          if (match_iou[j] < negative_mining_thresh && aflag[j] == -1) {
            /* .cc:218-232: softmax probability of background, float, sequential sum */
            float max_val = pred[j];
            for (int k = 1; k < Cp1; ++k) {
              float t = pred[j + (size_t)A * k];
              if (t > max_val) max_val = t;
            }
            float sum = 0.f;
            for (int k = 0; k < Cp1; ++k) sum += expf(pred[j + (size_t)A * k] - max_val);
            float prob = expf(pred[j] - max_val) / sum;
            cand[nc].value = -prob; cand[nc].index = j; ++nc;
          }
        }
        if (nc < num_negative) { rc = -3; num_negative = nc; }   /* CHECK_GE .cc:236 */
        qsort(cand, nc, sizeof(sort_elem), cmp_desc);
        for (int i = 0; i < num_negative; ++i) aflag[cand[i].index] = 0;
      }
    } else {
      for (int j = 0; j < A; ++j) if (aflag[j] != 1) aflag[j] = 0;
    }

    /* .cc:251-281 write targets */
    float *lt = loc_target + (size_t)nb * A * 5;
    float *lm = loc_mask + (size_t)nb * A * 5;
    float *ct = cls_target + (size_t)nb * A;
    for (int j = 0; j < A; ++j) {
      if (aflag[j] == 1) {
        const float *g = lab + label_w * match_gt[j];
        ct[j] = g[0] + 1;
        for (int q = 0; q < 5; ++q) lm[j * 5 + q] = 1;
        encode_loc(anchors + 4 * j, g + 1, lt + j * 5,
                   variances[0], variances[1], variances[2], variances[3]);
      } else if (aflag[j] == 0) {
        ct[j] = 0;
        for (int q = 0; q < 5; ++q) lm[j * 5 + q] = 0;
      }
    }
  }
  free(overlaps); free(match_iou); free(match_gt); free(aflag); free(gflag); free(cand);
  return rc;
}

/* ------------------------------------------------------------------ */
/* MultiBoxDetection                                                   */
/* ------------------------------------------------------------------ */
/* operator/multibox_detection.cc:44-51 */
static float nms_iou(const float *a, const float *b) {
  float w = fmax_(0.f, fmin_(a[2], b[2]) - fmax_(a[0], b[0]));
  float h = fmax_(0.f, fmin_(a[3], b[3]) - fmax_(a[1], b[1]));
  float i = w * h;
  float u = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - i;
  return u <= 0.f ? 0.f : i / u;
}
static inline float clip01(float v) { return fmax_(0.f, fmin_(1.f, v)); }

/* operator/multibox_detection-inl.h:81-107 (out = -1) and
 * operator/multibox_detection.cc:54-169.
 * Note on `exp(pw * vw)` (.cc:113): with DType=float this file evaluates it as
 * expf() and the following `* aw / 2` in float; an MXNet build that resolves
 * the unqualified call to ::exp(double) would differ in the last ulp of the
 * box coordinates (never in which rows are produced). */
int dspn_oracle_multibox_detection(const float *cls_prob /* (B,Cp1,A) */,
                                   const float *loc_pred /* (B,A*5) */,
                                   const float *anchors  /* (A,4) */,
                                   int B, int A, int Cp1,
                                   float threshold, int clip,
                                   const float variances[4],
                                   float nms_threshold, int force_suppress,
                                   int nms_topk,
                                   float *out /* (B,A,7) */) {
  if (B <= 0 || A <= 0 || Cp1 <= 0) return -1;
  const float vx = variances[0], vy = variances[1], vw = variances[2], vh = variances[3];
  for (long i = 0; i < (long)B * A * 7; ++i) out[i] = -1.f;
  float *temp = (float *)malloc(sizeof(float) * (size_t)A * 7);
  sort_elem *sorter = (sort_elem *)malloc(sizeof(sort_elem) * A);
  for (int nb = 0; nb < B; ++nb) {
    const float *prob = cls_prob + (size_t)nb * Cp1 * A;
    const float *loc = loc_pred + (size_t)nb * A * 5;
    float *po = out + (size_t)nb * A * 7;
    int valid = 0;
    for (int i = 0; i < A; ++i) {
      float score = -1; int id = 0;
      for (int j = 1; j < Cp1; ++j) {
        float t = prob[(size_t)j * A + i];
        if (t > score) { score = t; id = j; }
      }
      if (id > 0 && score < threshold) id = 0;
      if (id > 0) {
        float *row = po + valid * 7;
        row[0] = (float)(id - 1);
        row[1] = score;
        const float *an = anchors + 4 * i; const float *p = loc + 5 * i;
        float al = an[0], at = an[1], ar = an[2], ab = an[3];
        float aw = ar - al, ah = ab - at;
        float ax = (al + ar) / 2.f, ay = (at + ab) / 2.f;
        float ox = p[0] * vx * aw + ax;
        float oy = p[1] * vy * ah + ay;
#ifdef DSPN_ORACLE_EXP_DOUBLE
        /* the other reading of `exp(pw * vw) * aw / 2` (.cc:113): ::exp(double), the product and the division in double,
         * rounded once on assignment -- built as libdspn_oracle_expd.so; tests/test_oracle_multibox.py shows that ids, row
         * order and suppression are the same under both readings and the coordinates agree to 1 ulp */
        float ow = (float)(exp((double)(p[2] * vw)) * (double)aw / 2);
        float oh = (float)(exp((double)(p[3] * vh)) * (double)ah / 2);
#else
        float ow = expf(p[2] * vw) * aw / 2;
        float oh = expf(p[3] * vh) * ah / 2;
#endif
        float oz = (float)((double)p[4] * 0.1);
        row[2] = clip ? clip01(ox - ow) : ox - ow;
        row[3] = clip ? clip01(oy - oh) : oy - oh;
        row[4] = clip ? clip01(ox + ow) : ox + ow;
        row[5] = clip ? clip01(oy + oh) : oy + oh;
        row[6] = clip ? clip01(oz) : oz;
        ++valid;
      }
    }
    if (valid < 1 || nms_threshold <= 0 || nms_threshold > 1) continue;
    memcpy(temp, po, sizeof(float) * (size_t)A * 7);
    for (int i = 0; i < valid; ++i) { sorter[i].value = po[i * 7 + 1]; sorter[i].index = i; }
    qsort(sorter, valid, sizeof(sort_elem), cmp_desc);
    int nkeep = valid;
    if (nms_topk > 0 && nms_topk < nkeep) nkeep = nms_topk;
    /* only the first nkeep sorted rows are written back; rows nkeep..valid-1
     * keep their pre-sort content (.cc:143-151) */
    for (int i = 0; i < nkeep; ++i)
      memcpy(po + i * 7, temp + sorter[i].index * 7, sizeof(float) * 7);
    /* NMS runs over all `valid` rows (.cc:153-167) */
    for (int i = 0; i < valid; ++i) {
      float *ri = po + i * 7;
      if (ri[0] < 0) continue;
      for (int j = i + 1; j < valid; ++j) {
        float *rj = po + j * 7;
        if (rj[0] < 0) continue;
        if (force_suppress || ri[0] == rj[0]) {
          if (nms_iou(ri + 2, rj + 2) >= nms_threshold) rj[0] = -1;
        }
      }
    }
  }
  free(temp); free(sorter);
  return 0;
}
