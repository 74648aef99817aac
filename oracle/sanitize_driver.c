/* Host sanitizer run of the C oracle (test infrastructure, SURVEY.md section 5 "race detection / sanitizers": -fsanitize=address
 * on the host build).  Built by `make -C oracle sanitize` with -fsanitize=address,undefined and linked statically with
 * multibox_oracle.c; runs the three operators on exactly-sized heap buffers (so that any read or write past an array is a
 * report, not luck): random boxes, the degenerate inputs the GPU tests use (no labels, all-padding labels, A == 1, L == 1,
 * duplicate ground truths, every anchor identical, nms_topk larger / smaller than A), and returns 0 when nothing fired. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int dspn_oracle_multibox_prior(const float *sizes, int num_sizes, const float *ratios, int num_ratios, int in_height, int in_width,
                               float step_y, float step_x, float off_y, float off_x, int clip, float *out);
int dspn_oracle_multibox_target(const float *anchors, const float *labels, const float *cls_preds, int B, int A, int L, int label_w,
                                int Cp1, float overlap_threshold, float ignore_label, float negative_mining_ratio,
                                float negative_mining_thresh, int minimum_negative_samples, const float variances[4],
                                float *loc_target, float *loc_mask, float *cls_target);
int dspn_oracle_multibox_detection(const float *cls_prob, const float *loc_pred, const float *anchors, int B, int A, int Cp1,
                                   float threshold, int clip, const float variances[4], float nms_threshold, int force_suppress,
                                   int nms_topk, float *out);

static unsigned long long state = 88172645463325252ull;
static float urand(void) {   /* xorshift64, [0, 1) */
  state ^= state << 13; state ^= state >> 7; state ^= state << 17;
  return (float)((state >> 40) & 0xffffff) / 16777216.0f;
}

static int run_case(int B, int H, int W, int L, int Cp1, int mode, int topk) {
  const float sizes[2] = {0.2f, 0.272f}, ratios[3] = {1.f, 2.f, 0.5f};
  const float var[4] = {0.1f, 0.1f, 0.2f, 0.2f};
  const int per = 2 + 3 - 1, A = H * W * per;
  float *anchors = malloc(sizeof(float) * (size_t)A * 4);
  int bad = dspn_oracle_multibox_prior(sizes, 2, ratios, 3, H, W, -1.f, -1.f, 0.5f, 0.5f, mode == 3, anchors) != 0;
  if (mode == 4) for (int i = 1; i < A; ++i) memcpy(anchors + 4 * i, anchors, 16);   /* every anchor identical */
  float *labels = malloc(sizeof(float) * (size_t)B * L * 6);
  for (int b = 0; b < B; ++b)
    for (int l = 0; l < L; ++l) {
      float *r = labels + ((size_t)b * L + l) * 6;
      const int pad = mode == 1 || (mode == 0 && l >= L / 2) || (mode == 2 && b == 0);
      const float x0 = urand() * 0.7f, y0 = urand() * 0.7f;
      r[0] = pad ? -1.f : (float)((int)(urand() * (Cp1 - 1)));
      r[1] = x0; r[2] = y0; r[3] = x0 + 0.05f + urand() * 0.25f; r[4] = y0 + 0.05f + urand() * 0.25f; r[5] = urand() * 100.f;
      if (pad) r[1] = r[2] = r[3] = r[4] = r[5] = -1.f;      /* the padding rows the reference CHECKs for (.cc:99-103) */
      if (mode == 5 && l > 0) memcpy(r, r - 6, 24);                                    /* duplicate ground truths */
    }
  float *cls = malloc(sizeof(float) * (size_t)B * Cp1 * A), *loc = malloc(sizeof(float) * (size_t)B * A * 5);
  for (size_t i = 0; i < (size_t)B * Cp1 * A; ++i) cls[i] = urand();
  for (size_t i = 0; i < (size_t)B * A * 5; ++i) loc[i] = urand() - 0.5f;
  float *lt = malloc(sizeof(float) * (size_t)B * A * 5), *lm = malloc(sizeof(float) * (size_t)B * A * 5);
  float *ct = malloc(sizeof(float) * (size_t)B * A), *det = malloc(sizeof(float) * (size_t)B * A * 7);
  /* -2 / -3 are the reference's CHECK failures recorded as codes (the operator still fills its outputs); -1 = bad shapes */
  int rc = dspn_oracle_multibox_target(anchors, labels, cls, B, A, L, 6, Cp1, 0.5f, -1.f, 3.f, 0.5f, 0, var, lt, lm, ct);
  bad |= rc == -1;
  rc = dspn_oracle_multibox_detection(cls, loc, anchors, B, A, Cp1, 0.01f, 1, var, 0.5f, mode & 1, topk, det);
  bad |= rc == -1;
  double sum = 0;
  for (size_t i = 0; i < (size_t)B * A * 7; ++i) sum += det[i];
  for (size_t i = 0; i < (size_t)B * A; ++i) sum += ct[i];
  free(anchors); free(labels); free(cls); free(loc); free(lt); free(lm); free(ct); free(det);
  return bad | (sum != sum);   /* NaN would be a bug of its own */
}

int main(void) {
  int bad = 0;
  for (int mode = 0; mode <= 5; ++mode) {
    bad |= run_case(2, 8, 8, 6, 9, mode, 400);
    bad |= run_case(1, 1, 1, 1, 2, mode, 1);        /* A = 4, L = 1, one class */
    bad |= run_case(3, 5, 7, 3, 21, mode, 10);      /* nms_topk < A */
    bad |= run_case(1, 16, 16, 40, 9, mode, -1);    /* no top-k limit */
  }
  printf("sanitize_driver: %s\n", bad ? "FAILED" : "ok");
  return bad;
}
