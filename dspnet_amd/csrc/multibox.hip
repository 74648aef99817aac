// SSD multibox operators for MI355X (gfx950): MultiBoxPrior / MultiBoxTarget /
// MultiBoxDetection.  C ABI in include/dspn_multibox.h.
//
// These are HBM/latency-bound integer+float bookkeeping ops: no
// MFMA.  The design points are
//   * IoU is recomputed on the fly from the (A,4) anchor table (L2 resident)
//     and the <=L valid ground-truth boxes held in LDS, instead of the
//     reference's (11,B,A,L) float scratch (operator/multibox_target-inl.h:114-161);
//   * every data-dependent choice (bipartite match, hard-negative ranking, score
//     sort, NMS) is made with a TOTAL order identical to the CPU reference's
//     scan order / std::stable_sort, so indices are bit-exact and run-to-run
//     deterministic (the reference's CUDA path is neither);
//   * float arithmetic is kept bit-identical to the CPU reference: no FMA
//     contraction, IEEE divide/sqrt, and expf evaluated with glibc's own
//     algorithm (see expf_cr).
#include "dspn_common.h"
#include "../../include/dspn_multibox.h"
#include <cstdint>
#include <algorithm>

#pragma clang fp contract(off)

namespace {

constexpr int kMaxAttr = 32;     // max sizes / ratios per prior layer
constexpr int kMaxLabels = 1024; // max padded label rows per sample
constexpr int kTB = 1024;        // threads of the per-sample kernels

// expf with the bits of glibc's expf (sysdeps/ieee754/flt-32/e_expf.c: the
// table-of-32 2^(i/32) * cubic algorithm evaluated in double, FMA-contracted as
// the x86-64 FMA ifunc variant is).  glibc's expf is NOT correctly rounded
// (0.502 ulp; ~6e-4 of inputs differ from a rounded double exp), and the
// ranking of hard negatives / the NMS decisions depend on its exact bits, so
// the algorithm is replicated instead of approximated: on 2e8 random inputs the
// host model of this function matched glibc 2.35 bit for bit.
__constant__ unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

__device__ __forceinline__ float expf_cr(float x) {
  const double kInvLn2N = 0x1.71547652b82fep+0 * 32;
  const double kShift = 0x1.8p+52;
  const double c0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, c1 = 0x1.ebfce50fac4f3p-3 / 32 / 32,
               c2 = 0x1.62e42ff0c52d6p-1 / 32;
  if (!(fabsf(x) < 88.0f)) {
    if (x != x) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();
    if (x < -0x1.9fe368p6f) return 0.0f;
  }
  double z = kInvLn2N * (double)x;
  double kd = z + kShift;
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= kShift;
  const double r = z - kd;
  const unsigned long long t = kExp2fTab[ki & 31u] + (ki << 47);
  const double s = __longlong_as_double((long long)t);
  z = fma(c0, r, c1);
  const double r2 = r * r;
  double y = fma(c2, r, 1.0);
  y = fma(z, r2, y);
  y = y * s;
  return (float)y;
}
// glibc's logf is 0.818 ulp (table-driven, not reproducible without its table);
// it only feeds loc_target VALUES, never an index, so a correctly rounded log is
// used and the tests allow 1 ulp on those two columns.
__device__ __forceinline__ float logf_cr(float x) { return (float)log((double)x); }
__device__ __forceinline__ float fmaxr(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float fminr(float a, float b) { return a < b ? a : b; }

// ---------------------------------------------------------------------------
// MultiBoxPrior: one thread per (location, anchor slot)
// ---------------------------------------------------------------------------
struct PriorAttr {
  float sizes[kMaxAttr];
  float ratios[kMaxAttr];
};

__global__ void prior_kernel(PriorAttr attr, int num_sizes, int num_ratios, int H, int W,
                             float step_y, float step_x, float off_y, float off_x, int clip,
                             float4 *__restrict__ out) {
  const int per = num_sizes + num_ratios - 1;
  const int total = H * W * per;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int loc = idx / per, slot = idx - loc * per;
  const int r = loc / W, c = loc - r * W;
  const float cy = (r + off_y) * step_y;
  const float cx = (c + off_x) * step_x;
  float w, h;
  if (slot < num_sizes) {
    const float size = attr.sizes[slot];
    w = size * H / W / 2;
    h = size / 2;
  } else {
    const float size = attr.sizes[0];
    const float ratio = sqrtf(attr.ratios[slot - num_sizes + 1]);
    w = size * H / W * ratio / 2;
    h = size / ratio / 2;
  }
  float4 o = make_float4(cx - w, cy - h, cx + w, cy + h);
  if (clip) {
    o.x = o.x < 0.f ? 0.f : (o.x > 1.f ? 1.f : o.x);
    o.y = o.y < 0.f ? 0.f : (o.y > 1.f ? 1.f : o.y);
    o.z = o.z < 0.f ? 0.f : (o.z > 1.f ? 1.f : o.z);
    o.w = o.w < 0.f ? 0.f : (o.w > 1.f ? 1.f : o.w);
  }
  out[idx] = o;
}

// ---------------------------------------------------------------------------
// MultiBoxTarget
// ---------------------------------------------------------------------------
__device__ __forceinline__ float target_iou(const float4 a, const float gl, const float gt,
                                            const float gr, const float gb) {
  const float iw = fmaxr(0.f, fminr(a.z, gr) - fmaxr(a.x, gl));
  const float ih = fmaxr(0.f, fminr(a.w, gb) - fmaxr(a.y, gt));
  const float inter = iw * ih;
  const float uni = (a.z - a.x) * (a.w - a.y) + (gr - gl) * (gb - gt) - inter;
  return uni == 0.f ? 0.f : inter / uni;
}

struct TargetWs {
  int *err;            // [B]
  int *ngt;            // [B]
  float *row_iou;      // [B*A] best IoU of each anchor over the valid GTs
  int *row_gt;         // [B*A] arg of it (later: matched GT of positives)
  unsigned *bgkey;     // [B*A] float bits of softmax P(background)
  signed char *flag;   // [B*A] -1 ignore, 0 negative, 1 positive
  unsigned long long *cpart;   // [B*nblk*L] per (256-anchor block, GT): best anchor of the block, packed (IoU bits, ~anchor); 0 = none
  int nblk;            // anchor blocks per sample
};

// number of valid GT rows = index of the first row whose class is -1
__device__ int count_valid_gt(const float *lab, int L, int lw, int *s_G) {
  if (threadIdx.x == 0) *s_G = L;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x)
    if (lab[i * lw] == -1.0f) atomicMin(s_G, i);
  __syncthreads();
  return *s_G;
}

// largest value of the wave, in every lane (row reductions by DPP, then across the four rows)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  auto step = [](unsigned x, unsigned y) { return x > y ? x : y; };
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
  v = step(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));   // row_mirror: every lane holds its row's max
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return step(step(a, b), step(c, d));
}

// K1: per anchor, best GT by IoU and softmax background probability; per GT, the best anchor of this 256-anchor block.
// Round 5: the column maxima (best anchor of every GT: G x A IoUs per sample) were the matching kernel's largest phase, on ONE
// compute unit per sample; every one of those IoUs is formed here anyway, on the whole chip.  A block's candidate for GT k is
// packed as (IoU bits << 32 | ~anchor): unsigned order = larger IoU first, then the lower anchor, the reference's scan order
// (multibox_target.cc:113-149); only IoU > 0 can ever be matched (the cut is 1e-6), so 0 stands for "none".
__global__ __launch_bounds__(256) void target_rows_kernel(
    const float4 *__restrict__ anchors, const float *__restrict__ labels,
    const float *__restrict__ cls_preds, int A, int L, int lw, int C, TargetWs ws) {
  __shared__ float s_gt[kMaxLabels * 4];
  __shared__ unsigned long long s_cmax[kMaxLabels];
  __shared__ int s_G;
  const int b = blockIdx.y;
  const float *lab = labels + (size_t)b * L * lw;
  const int G = count_valid_gt(lab, L, lw, &s_G);
  for (int i = threadIdx.x; i < G * 4; i += blockDim.x) {
    const int k = i >> 2, q = i & 3;
    s_gt[i] = lab[k * lw + 1 + q];
  }
  for (int k = threadIdx.x; k < G; k += blockDim.x) s_cmax[k] = 0ull;
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t o = (size_t)b * A + j;
  if (G == 0) {  // nothing downstream reads the workspace of an empty sample
    if (j < A) { ws.row_iou[o] = -1.f; ws.row_gt[o] = -1; ws.bgkey[o] = 0u; }
    return;
  }
  // (every lane of a wave takes part in the loop -- wave_max_u32 is a butterfly over all 64 lanes --; the lanes past the last
  // anchor work on a copy of it and contribute nothing)
  const bool live = j < A;
  const float4 a = anchors[live ? j : A - 1];
  float best = -1.0f; int bk = -1;
  for (int k = 0; k < G; ++k) {
    const float iou = target_iou(a, s_gt[4 * k], s_gt[4 * k + 1], s_gt[4 * k + 2], s_gt[4 * k + 3]);
    if (iou > best) { best = iou; bk = k; }
    // the wave's best for GT k: largest IoU bits (positive floats order as unsigned), lowest lane = lowest anchor among equals
    const unsigned bits = (live && iou > 0.f) ? __float_as_uint(iou) : 0u;
    const unsigned top = wave_max_u32(bits);
    if (top != 0u) {
      const int first = __ffsll((long long)__ballot(bits == top)) - 1;
      if ((threadIdx.x & 63) == first)
        atomicMax(&s_cmax[k], ((unsigned long long)top << 32) | (0xffffffffu - (unsigned)j));
    }
  }
  if (live) {
    ws.row_iou[o] = best;
    ws.row_gt[o] = bk;
    // softmax P(background), float, sequential sum (multibox_target.cc:218-232)
    const float *p = cls_preds + (size_t)b * C * A + j;
    const float p0 = p[0];
    float mx = p0;
    for (int k = 1; k < C; ++k) { const float t = p[(size_t)k * A]; if (t > mx) mx = t; }
    float sum = 0.f;
    for (int k = 0; k < C; ++k) sum += expf_cr(p[(size_t)k * A] - mx);
    const float prob = expf_cr(p0 - mx) / sum;
    ws.bgkey[o] = __float_as_uint(prob);
  }
  __syncthreads();
  unsigned long long *cp = ws.cpart + ((size_t)b * ws.nblk + blockIdx.x) * L;
  for (int k = threadIdx.x; k < G; k += blockDim.x) cp[k] = s_cmax[k];
}

struct Best { float iou; int a; int k; };
// order of the reference's scan (anchor-major, gt-minor, strict '>'):
// larger IoU first, then lower anchor, then lower gt.
__device__ __forceinline__ bool better(const Best &x, const Best &y) {
  if (x.iou != y.iou) return x.iou > y.iou;
  if (x.a != y.a) return x.a < y.a;
  return x.k < y.k;
}
__device__ __forceinline__ Best wave_best(Best v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    Best o;
    o.iou = __shfl_xor(v.iou, m, 64);
    o.a = __shfl_xor(v.a, m, 64);
    o.k = __shfl_xor(v.k, m, 64);
    if (better(o, v)) v = o;
  }
  return v;
}

// exclusive scan of a predicate over the 1024-thread block; returns this
// thread's offset, sets total.  s_w: int[16].  Two barriers.
__device__ __forceinline__ int block_scan_pred(bool f, int *s_w, int &total) {
  const unsigned long long bal = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wprefix = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) s_w[wave] = __popcll(bal);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kTB / 64; ++w) { const int c = s_w[w]; if (w < wave) off += c; tot += c; }
  __syncthreads();
  total = tot;
  return off + wprefix;
}

// K2: one workgroup per sample: bipartite stage, threshold stage, mining.
__global__ __launch_bounds__(kTB) void target_match_kernel(
    const float4 *__restrict__ anchors, const float *__restrict__ labels, int A, int L, int lw,
    float overlap_threshold, float neg_ratio, float neg_thresh, TargetWs ws) {
  __shared__ float s_gt[kMaxLabels * 4];
  __shared__ float s_ciou[kMaxLabels];
  __shared__ int s_ca[kMaxLabels];
  __shared__ int s_gflag[kMaxLabels];
  __shared__ Best s_wb[kTB / 64];
  __shared__ Best s_best;
  __shared__ unsigned s_hist[256];
  __shared__ int s_w[kTB / 64];
  __shared__ int s_G, s_cnt, s_bin, s_kk;

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *lab = labels + (size_t)b * L * lw;
  const int G = count_valid_gt(lab, L, lw, &s_G);
  if (tid == 0) {
    int err = 0;
    if (G < L) {  // CHECK_EQ x4 on the terminating row (multibox_target.cc:98-101)
      const float *row = lab + G * lw;
      if (row[1] != -1.0f || row[2] != -1.0f || row[3] != -1.0f || row[4] != -1.0f) err = 2;
    }
    ws.err[b] = err;
    ws.ngt[b] = G;
  }
  if (G == 0) return;
  float *row_iou = ws.row_iou + (size_t)b * A;
  int *row_gt = ws.row_gt + (size_t)b * A;
  unsigned *bgkey = ws.bgkey + (size_t)b * A;
  signed char *flag = ws.flag + (size_t)b * A;

  for (int i = tid; i < G * 4; i += kTB) s_gt[i] = lab[(i >> 2) * lw + 1 + (i & 3)];
  for (int j = tid; j < A; j += kTB) flag[j] = -1;
  __syncthreads();

  // column maxima: best unmatched anchor of every GT, one wave per GT
  for (int k = wave; k < G; k += kTB / 64) {
    const float gl = s_gt[4 * k], gt = s_gt[4 * k + 1], gr = s_gt[4 * k + 2], gb = s_gt[4 * k + 3];
    Best v{-1.0f, 0x7fffffff, k};
    for (int j = lane; j < A; j += 64) {
      const float iou = target_iou(anchors[j], gl, gt, gr, gb);
      if (iou > v.iou) { v.iou = iou; v.a = j; }
    }
    v = wave_best(v);
    if (lane == 0) { s_ciou[k] = v.iou; s_ca[k] = v.a; s_gflag[k] = 0; }
  }
  __syncthreads();

  // greedy bipartite matching (multibox_target.cc:113-149).  Round 4: wave 0 alone runs consecutive picks -- arg-max over
  // the unmatched GTs in the total order (IoU desc, anchor asc, gt asc), mark, next -- and the workgroup only meets at a
  // barrier when a pick consumed an anchor that is still some unmatched GT's column maximum, which then has to be recomputed
  // by all 1024 threads (rare: two ground truths sharing their best anchor).  The picks and their order are exactly those
  // of the one-pick-per-barrier loop this replaces (3 barriers x up to 40 picks per sample: ~0.2 ms of a 0.30 ms kernel).
  int npos = 0;
  for (;;) {
    if (wave == 0) {
      int consumed = -1;                 // anchor whose consumption forces a recompute (-1: matching is complete)
      for (;;) {
        Best v{-2.0f, 0x7fffffff, -1};
        for (int k = lane; k < G; k += 64) {
          if (s_gflag[k]) continue;
          Best c{s_ciou[k], s_ca[k], k};
          if (better(c, v)) v = c;
        }
        v = wave_best(v);
        if (v.k < 0 || !(v.iou > 1e-6f)) break;
        if (lane == 0) { flag[v.a] = 1; row_gt[v.a] = v.k; s_gflag[v.k] = 1; }
        ++npos;
        // (the LDS accesses of one wave complete in order; the fence keeps hipcc from carrying s_gflag over in registers)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        bool hit = false;
        for (int k = lane; k < G; k += 64) hit |= (k != v.k && !s_gflag[k] && s_ca[k] == v.a);
        if (__ballot(hit) != 0ull) { consumed = v.a; break; }
      }
      if (lane == 0) { s_best.a = consumed; s_best.k = npos; }
    }
    __syncthreads();
    const int consumed = s_best.a;
    npos = s_best.k;
    if (consumed < 0) break;
    // GTs whose best anchor was just consumed need a new column maximum
    for (int k = 0; k < G; ++k) {
      if (s_gflag[k] || s_ca[k] != consumed) continue;   // block-uniform
      const float gl = s_gt[4 * k], gt = s_gt[4 * k + 1], gr = s_gt[4 * k + 2], gb = s_gt[4 * k + 3];
      Best v{-1.0f, 0x7fffffff, k};
      for (int j = tid; j < A; j += kTB) {
        if (flag[j] == 1) continue;
        const float iou = target_iou(anchors[j], gl, gt, gr, gb);
        if (iou > v.iou) { v.iou = iou; v.a = j; }
      }
      v = wave_best(v);
      if (lane == 0) s_wb[wave] = v;
      __syncthreads();
      if (wave == 0) {
        Best u = lane < kTB / 64 ? s_wb[lane] : Best{-1.0f, 0x7fffffff, k};
        u = wave_best(u);
        if (lane == 0) { s_ciou[k] = u.iou; s_ca[k] = u.a; }
      }
      __syncthreads();
    }
  }
  __syncthreads();

  // threshold stage (multibox_target.cc:151-180) + candidate census
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  {
    int add = 0;
    if (overlap_threshold > 0) {
      for (int j = tid; j < A; j += kTB) {
        if (flag[j] == 1) continue;
        if (row_iou[j] > overlap_threshold) { flag[j] = 1; ++add; }
      }
    }
    for (int m = 32; m >= 1; m >>= 1) add += __shfl_xor(add, m, 64);
    if (lane == 0 && add) atomicAdd(&s_cnt, add);
  }
  __syncthreads();
  npos += s_cnt;
  __syncthreads();

  if (!(neg_ratio > 0)) {  // use all negatives (multibox_target.cc:242-249)
    for (int j = tid; j < A; j += kTB) if (flag[j] != 1) flag[j] = 0;
    return;
  }
  int num_negative = (int)((float)npos * neg_ratio);
  if (num_negative > A - npos) num_negative = A - npos;
  if (num_negative <= 0) return;

  // candidates: not positive and best IoU below the mining threshold
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  {
    int c = 0;
    for (int j = tid; j < A; j += kTB) c += (flag[j] != 1 && row_iou[j] < neg_thresh) ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if (lane == 0 && c) atomicAdd(&s_cnt, c);
  }
  __syncthreads();
  const int ncand = s_cnt;
  if (ncand < num_negative) {  // CHECK_GE(temp.size(), num_negative), multibox_target.cc:236
    if (tid == 0) ws.err[b] = 3;
    num_negative = ncand;
  }
  if (num_negative == ncand) {
    for (int j = tid; j < A; j += kTB)
      if (flag[j] != 1 && row_iou[j] < neg_thresh) flag[j] = 0;
    return;
  }

  // radix-select the num_negative-th smallest P(background); ties go to the
  // lower anchor index ( == std::stable_sort on -prob, multibox_target.cc:58-70,237 )
  unsigned prefix = 0;
  int kk = num_negative;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += kTB) s_hist[i] = 0;
    __syncthreads();
    for (int j = tid; j < A; j += kTB) {
      if (flag[j] == 1 || !(row_iou[j] < neg_thresh)) continue;
      const unsigned key = bgkey[j];
      const bool match = (shift == 24) ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
      if (match) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int cum = 0, bin = 0;
      for (; bin < 256; ++bin) {
        const int h = (int)s_hist[bin];
        if (cum + h >= kk) break;
        cum += h;
      }
      s_bin = bin; s_kk = kk - cum;
    }
    __syncthreads();
    prefix |= (unsigned)s_bin << shift;
    kk = s_kk;
    __syncthreads();
  }
  const unsigned T = prefix;
  int taken_ties = 0;
  for (int base = 0; base < A; base += kTB) {
    const int j = base + tid;
    bool cand = false; unsigned key = 0;
    if (j < A) { cand = (flag[j] != 1 && row_iou[j] < neg_thresh); key = bgkey[j]; }
    const bool tie = cand && key == T;
    int total;
    const int off = block_scan_pred(tie, s_w, total);
    if (cand && (key < T || (tie && taken_ties + off < kk))) flag[j] = 0;
    taken_ties += total;
  }
}

// K2, A <= kPer x 1024 anchors (round 5): the same stages with every per-anchor quantity where one compute unit can reach it
// without a trip to L2 -- best IoU and background key of a thread's kPer anchors in registers, the flags in LDS -- and the
// column maxima merged from target_rows_kernel's per-block partials instead of G x A IoUs formed here.  The generic kernel
// above made some ten sweeps over global arrays, six dependent load rounds each (0.17 ms at 32 x 6132, of which the column
// maxima were about half).  Decisions and their order are those of the generic kernel.
template <int kPer>
__global__ __launch_bounds__(kTB) void target_match_reg_kernel(
    const float4 *__restrict__ anchors, const float *__restrict__ labels, int A, int L, int lw,
    float overlap_threshold, float neg_ratio, float neg_thresh, TargetWs ws) {
  __shared__ float s_gt[kMaxLabels * 4];
  __shared__ float s_ciou[kMaxLabels];
  __shared__ int s_ca[kMaxLabels];
  __shared__ int s_gflag[kMaxLabels];
  __shared__ Best s_wb[kTB / 64];
  __shared__ Best s_best;
  __shared__ unsigned s_hist[256];
  __shared__ int s_w[kTB / 64];
  __shared__ int s_G, s_cnt, s_bin, s_kk, s_ties, s_dup;
  __shared__ signed char s_flag[kPer * kTB];
  __shared__ unsigned long long s_cmax[kMaxLabels];
  __shared__ unsigned s_used[kPer * kTB / 32];      // anchors that are some matchable GT's column maximum

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *lab = labels + (size_t)b * L * lw;
  const int G = count_valid_gt(lab, L, lw, &s_G);
  if (tid == 0) {
    int err = 0;
    if (G < L) {  // CHECK_EQ x4 on the terminating row (multibox_target.cc:98-101)
      const float *row = lab + G * lw;
      if (row[1] != -1.0f || row[2] != -1.0f || row[3] != -1.0f || row[4] != -1.0f) err = 2;
    }
    ws.err[b] = err;
    ws.ngt[b] = G;
  }
  if (G == 0) return;
  int *row_gt = ws.row_gt + (size_t)b * A;
  signed char *flag = ws.flag + (size_t)b * A;

  float riou[kPer];
  unsigned key[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int j = u * kTB + tid;
    riou[u] = j < A ? ws.row_iou[(size_t)b * A + j] : 0.f;
    key[u] = j < A ? ws.bgkey[(size_t)b * A + j] : 0u;
    s_flag[j] = -1;
  }
  auto flush = [&]() {      // a thread's own anchors: nobody else writes them after the matching stage
#pragma unroll
    for (int u = 0; u < kPer; ++u) { const int j = u * kTB + tid; if (j < A) flag[j] = s_flag[j]; }
  };
  for (int i = tid; i < G * 4; i += kTB) s_gt[i] = lab[(i >> 2) * lw + 1 + (i & 3)];
  // column maxima: the blocks' partials, one (GT, block) pair per thread
  for (int k = tid; k < G; k += kTB) s_cmax[k] = 0ull;
  for (int i = tid; i < kPer * kTB / 32; i += kTB) s_used[i] = 0u;
  if (tid == 0) s_dup = 0;
  __syncthreads();
  for (int i = tid; i < G * ws.nblk; i += kTB) {
    const int q = i / G, k = i - q * G;
    const unsigned long long c = ws.cpart[((size_t)b * ws.nblk + q) * L + k];
    if (c) atomicMax(&s_cmax[k], c);
  }
  __syncthreads();
  for (int k = tid; k < G; k += kTB) {
    const unsigned long long m = s_cmax[k];
    const float ciou = m ? __uint_as_float((unsigned)(m >> 32)) : -1.0f;
    const int ca = m ? (int)(0xffffffffu - (unsigned)m) : 0x7fffffff;
    s_ciou[k] = ciou; s_ca[k] = ca; s_gflag[k] = 0;
    if (ciou > 1e-6f) {       // would be picked: does another such GT share its best anchor?
      const unsigned bit = 1u << (ca & 31);
      if (atomicOr(&s_used[ca >> 5], bit) & bit) s_dup = 1;
    }
  }
  __syncthreads();

  // greedy bipartite matching (multibox_target.cc:113-149), as in the generic kernel: wave 0 runs consecutive picks, the
  // workgroup meets only when a pick consumed an anchor that is still some unmatched GT's column maximum
  int npos = 0;
  if (!s_dup) {
    // No two ground truths that can be matched (IoU > 1e-6) share their best anchor: every pick of the greedy loop takes a GT
    // with its column maximum, no pick changes another GT's maximum that matters (a GT at or below the cut may lose its
    // anchor, but whatever replaces it is no larger and never picked) -- the result is all of them at once.
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int add = 0;
    for (int k = tid; k < G; k += kTB)
      if (s_ciou[k] > 1e-6f) { s_flag[s_ca[k]] = 1; row_gt[s_ca[k]] = k; ++add; }
    for (int m = 32; m >= 1; m >>= 1) add += __shfl_xor(add, m, 64);
    if (lane == 0 && add) atomicAdd(&s_cnt, add);
    __syncthreads();
    npos = s_cnt;
  } else
  for (;;) {
    if (wave == 0) {
      int consumed = -1;
      for (;;) {
        Best v{-2.0f, 0x7fffffff, -1};
        for (int k = lane; k < G; k += 64) {
          if (s_gflag[k]) continue;
          Best c{s_ciou[k], s_ca[k], k};
          if (better(c, v)) v = c;
        }
        v = wave_best(v);
        if (v.k < 0 || !(v.iou > 1e-6f)) break;
        if (lane == 0) { s_flag[v.a] = 1; row_gt[v.a] = v.k; s_gflag[v.k] = 1; }
        ++npos;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        bool hit = false;
        for (int k = lane; k < G; k += 64) hit |= (k != v.k && !s_gflag[k] && s_ca[k] == v.a);
        if (__ballot(hit) != 0ull) { consumed = v.a; break; }
      }
      if (lane == 0) { s_best.a = consumed; s_best.k = npos; }
    }
    __syncthreads();
    const int consumed = s_best.a;
    npos = s_best.k;
    if (consumed < 0) break;
    for (int k = 0; k < G; ++k) {
      if (s_gflag[k] || s_ca[k] != consumed) continue;   // block-uniform
      const float gl = s_gt[4 * k], gt = s_gt[4 * k + 1], gr = s_gt[4 * k + 2], gb = s_gt[4 * k + 3];
      Best v{-1.0f, 0x7fffffff, k};
      for (int j = tid; j < A; j += kTB) {
        if (s_flag[j] == 1) continue;
        const float iou = target_iou(anchors[j], gl, gt, gr, gb);
        if (iou > v.iou) { v.iou = iou; v.a = j; }
      }
      v = wave_best(v);
      if (lane == 0) s_wb[wave] = v;
      __syncthreads();
      if (wave == 0) {
        Best u = lane < kTB / 64 ? s_wb[lane] : Best{-1.0f, 0x7fffffff, k};
        u = wave_best(u);
        if (lane == 0) { s_ciou[k] = u.iou; s_ca[k] = u.a; }
      }
      __syncthreads();
    }
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();

  // threshold stage (multibox_target.cc:151-180)
  {
    int add = 0;
    if (overlap_threshold > 0) {
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const int j = u * kTB + tid;
        if (j < A && s_flag[j] != 1 && riou[u] > overlap_threshold) { s_flag[j] = 1; ++add; }
      }
    }
    for (int m = 32; m >= 1; m >>= 1) add += __shfl_xor(add, m, 64);
    if (lane == 0 && add) atomicAdd(&s_cnt, add);
  }
  __syncthreads();
  npos += s_cnt;
  __syncthreads();

  if (!(neg_ratio > 0)) {  // use all negatives (multibox_target.cc:242-249)
#pragma unroll
    for (int u = 0; u < kPer; ++u) { const int j = u * kTB + tid; if (s_flag[j] != 1) s_flag[j] = 0; }
    flush();
    return;
  }
  int num_negative = (int)((float)npos * neg_ratio);
  if (num_negative > A - npos) num_negative = A - npos;
  if (num_negative <= 0) { flush(); return; }

  // candidates: not positive and best IoU below the mining threshold
  unsigned cand = 0;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  {
    int c = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int j = u * kTB + tid;
      if (j < A && s_flag[j] != 1 && riou[u] < neg_thresh) { cand |= 1u << u; ++c; }
    }
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if (lane == 0 && c) atomicAdd(&s_cnt, c);
  }
  __syncthreads();
  const int ncand = s_cnt;
  if (ncand < num_negative) {  // CHECK_GE(temp.size(), num_negative), multibox_target.cc:236
    if (tid == 0) ws.err[b] = 3;
    num_negative = ncand;
  }
  if (num_negative == ncand) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) if ((cand >> u) & 1u) s_flag[u * kTB + tid] = 0;
    flush();
    return;
  }

  // radix-select the num_negative-th smallest P(background); ties go to the lower anchor index
  // ( == std::stable_sort on -prob, multibox_target.cc:58-70,237 )
  unsigned prefix = 0;
  int kk = num_negative;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += kTB) s_hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      if (!((cand >> u) & 1u)) continue;
      const bool match = (shift == 24) ? true : ((key[u] >> (shift + 8)) == (prefix >> (shift + 8)));
      if (match) atomicAdd(&s_hist[(key[u] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (wave == 0) {      // first bin whose running count reaches kk: four bins per lane, a shuffle scan over the lanes
      unsigned h[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) h[q] = s_hist[4 * lane + q];
      const int mine = (int)(h[0] + h[1] + h[2] + h[3]);
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
      const unsigned long long reach = __ballot(incl >= kk);
      const int first = __ffsll((long long)reach) - 1;     // (ncand >= kk candidates match the prefix: some lane reaches it)
      if (lane == first) {
        int cum = incl - mine, bin = 4 * lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (cum + (int)h[q] >= kk) { bin = 4 * lane + q; break; }
          cum += (int)h[q];
        }
        s_bin = bin; s_kk = kk - cum; s_ties = (int)s_hist[bin];
      }
    }
    __syncthreads();
    prefix |= (unsigned)s_bin << shift;
    kk = s_kk;
  }
  const unsigned T = prefix;
  if (kk == s_ties) {       // every anchor at the cut is taken: no order among them is needed (the usual case)
#pragma unroll
    for (int u = 0; u < kPer; ++u) if (((cand >> u) & 1u) && key[u] <= T) s_flag[u * kTB + tid] = 0;
    flush();
    return;
  }
  int taken_ties = 0;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const bool c = (cand >> u) & 1u;
    const bool tie = c && key[u] == T;
    int total;
    const int off = block_scan_pred(tie, s_w, total);
    if (c && (key[u] < T || (tie && taken_ties + off < kk))) s_flag[u * kTB + tid] = 0;
    taken_ties += total;
  }
  flush();
}

// K3: expand flags into the three outputs (multibox_target.cc:251-281; init
// values of multibox_target-inl.h:121-123 for untouched anchors).
__global__ __launch_bounds__(256) void target_write_kernel(
    const float4 *__restrict__ anchors, const float *__restrict__ labels, int A, int L, int lw,
    float ignore_label, float vx, float vy, float vw, float vh, TargetWs ws,
    float *__restrict__ loc_target, float *__restrict__ loc_mask, float *__restrict__ cls_target) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A) return;
  const size_t o = (size_t)b * A + j;
  const int G = ws.ngt[b];
  const int f = G > 0 ? (int)ws.flag[o] : -1;
  float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float m = 0.f, c = ignore_label;
  if (f == 1) {
    const float *g = labels + ((size_t)b * L + ws.row_gt[o]) * lw;
    c = g[0] + 1;
    m = 1.f;
    const float4 a = anchors[j];
    const float aw = a.z - a.x, ah = a.w - a.y;
    const float ax = (float)((a.x + a.z) * 0.5), ay = (float)((a.y + a.w) * 0.5);
    const float gl = g[1], gt = g[2], gr = g[3], gb = g[4], gz = g[5];
    const float gw = gr - gl, gh = gb - gt;
    const float gx = (float)((gl + gr) * 0.5), gy = (float)((gt + gb) * 0.5);
    t[0] = (gx - ax) / aw / vx;
    t[1] = (gy - ay) / ah / vy;
    t[2] = logf_cr(gw / aw) / vw;
    t[3] = logf_cr(gh / ah) / vh;
    t[4] = (float)((double)gz / 0.1);
  } else if (f == 0) {
    c = 0.f;
  }
  cls_target[o] = c;
  float *lt = loc_target + o * 5, *lm = loc_mask + o * 5;
#pragma unroll
  for (int q = 0; q < 5; ++q) { lt[q] = t[q]; lm[q] = m; }
}

// ---------------------------------------------------------------------------
// MultiBoxDetection
// ---------------------------------------------------------------------------
struct DetWs {
  int *nms_count;              // [B] rows that take part in sort+NMS (0 = none)
  float *temp;                 // [B*A*7] pre-sort copy of the compacted rows
  unsigned long long *keys;    // [B*n2] sort keys when they do not fit LDS
  unsigned long long *mask;    // [B*A*nwords] suppression bit matrix, over GROUPED positions (see det_decode_sort_kernel)
  int *perm;                   // [B*A] grouped position -> output row: the valid rows ordered by (class id, row)
  int *gcls;                   // [B*A] class id of each grouped position (ascending)
  float4 *gbox;                // [B*A*2] per grouped position: (x1, y1, x2, y2), (area, class id, 0, 0) -- what nms_mask_kernel reads
};

__device__ __forceinline__ unsigned ordered_bits(float s) {
  if (s == 0.f) s = 0.f;  // -0 -> +0
  const unsigned u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float clip01(float v) { return fmaxr(0.f, fminr(1.f, v)); }

// in-place ascending bitonic sort of n2 (a power of two) 64-bit keys by one workgroup; ends behind a barrier.
// Round 4: each of the 16 waves owns a contiguous segment of S = n2 / 16 keys; a pass whose partner distance j is below S
// pairs keys of ONE segment, so a wave runs all those passes of a merge on its own -- its LDS operations complete in order,
// a wavefront fence between passes is enough -- and the workgroup meets only for the passes with j >= S and once per merge:
// 8192 keys take 10 + 13 workgroup barriers instead of 91 (MultiBoxDetection at 32 x 6132 rows: 0.34 -> see DESIGN.md).
template <bool kLds>      // keys in LDS (else in global memory: every pass by the workgroup, as before)
__device__ __forceinline__ void bitonic_sort_keys(unsigned long long *keys, int n2, int tid) {
  // pair p of a pass with partner distance j: i = p with a zero inserted at bit log2(j), partner i | j -- every thread of a
  // pass works (no idle half), and its (up to) four pairs are all loaded before the first is compared and stored: four
  // independent LDS round trips per pass instead of eight dependent ones
  auto pass = [&](const int p0, const int p1, const int step, const int j, const int k) __attribute__((always_inline)) {
    for (int p = p0; p < p1; p += 4 * step) {
      unsigned long long x[4], y[4];
      int a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = p + u * step;
        a[u] = q < p1 ? (((q & ~(j - 1)) << 1) | (q & (j - 1))) : -1;
        if (a[u] >= 0) { x[u] = keys[a[u]]; y[u] = keys[a[u] | j]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (a[u] < 0) continue;
        const bool asc = (a[u] & k) == 0;
        if ((x[u] > y[u]) == asc) { keys[a[u]] = y[u]; keys[a[u] | j] = x[u]; }
      }
    }
  };
  constexpr int kWaves = kTB / 64;
  const int S = (kLds && n2 >= 128 * kWaves) ? n2 / kWaves : 0;      // segment per wave (>= 128 keys), 0: every pass by the workgroup
  const int wave = tid >> 6, lane = tid & 63;
  if (kLds && n2 <= 1024) {     // round 5: a short list (the nms_topk rows) is sorted by ONE wave, no workgroup barrier per pass
    if (wave == 0) {
      for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          pass(lane, n2 >> 1, 64, j, k);
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    return;
  }
  for (int k = 2; k <= n2; k <<= 1) {
    int j = k >> 1;
    for (; j > 0 && j >= S; j >>= 1) {
      pass(tid, n2 >> 1, kTB, j, k);
      __syncthreads();
    }
    if (j > 0) {       // (S > 0) the remaining passes of this merge stay inside the waves' segments (pairs wave * S / 2 ...)
      const int lo = wave * (S >> 1);
      for (; j > 0; j >>= 1) {
        pass(lo + lane, lo + (S >> 1), 64, j, k);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      __syncthreads();
    }
  }
}

constexpr int kCountCls = 256;   // classes the counting sort of the grouping step holds in LDS
template <bool kLdsKeys>
__global__ __launch_bounds__(kTB) void det_decode_sort_kernel(
    const float *__restrict__ cls_prob, const float *__restrict__ loc_pred,
    const float4 *__restrict__ anchors, int A, int C, float threshold, int clip,
    float vx, float vy, float vw, float vh, int nms_enabled, int nms_topk, int force, int n2cap, int selcap,
    DetWs ws, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_w[kTB / 64];
  __shared__ unsigned s_hist[256];
  __shared__ int s_sel[4];                       // radix select: bin, remaining rank, ties in the bin, slot counter
  __shared__ int s_ccnt[kTB / 64][kCountCls];    // grouping: rows of a class in a wave's range, then its running offset
  __shared__ int s_ctot[kCountCls];
  const int b = blockIdx.x, tid = threadIdx.x;
  unsigned long long *keys = kLdsKeys ? reinterpret_cast<unsigned long long *>(smem)
                                      : ws.keys + (size_t)b * n2cap;
  const float *prob = cls_prob + (size_t)b * C * A;
  const float *loc = loc_pred + (size_t)b * A * 5;
  float *po = out + (size_t)b * A * 7;
  float *pt = ws.temp + (size_t)b * A * 7;

  int V = 0;
  for (int base = 0; base < A; base += kTB) {
    const int i = base + tid;
    float score = -1.f; int id = 0;
    if (i < A) {
      for (int j = 1; j < C; ++j) {
        const float t = prob[(size_t)j * A + i];
        if (t > score) { score = t; id = j; }
      }
      if (id > 0 && score < threshold) id = 0;
    }
    int total;
    const int pos = V + block_scan_pred(id > 0, s_w, total);
    if (id > 0) {
      const float4 a = anchors[i];
      const float *p = loc + (size_t)i * 5;
      const float aw = a.z - a.x, ah = a.w - a.y;
      const float ax = (a.x + a.z) / 2.f, ay = (a.y + a.w) / 2.f;
      const float ox = p[0] * vx * aw + ax;
      const float oy = p[1] * vy * ah + ay;
      const float ow = expf_cr(p[2] * vw) * aw / 2;
      const float oh = expf_cr(p[3] * vh) * ah / 2;
      const float oz = (float)((double)p[4] * 0.1);
      float r[7];
      r[0] = (float)(id - 1);
      r[1] = score;
      r[2] = clip ? clip01(ox - ow) : ox - ow;
      r[3] = clip ? clip01(oy - oh) : oy - oh;
      r[4] = clip ? clip01(ox + ow) : ox + ow;
      r[5] = clip ? clip01(oy + oh) : oy + oh;
      r[6] = clip ? clip01(oz) : oz;
#pragma unroll
      for (int q = 0; q < 7; ++q) { po[(size_t)pos * 7 + q] = r[q]; pt[(size_t)pos * 7 + q] = r[q]; }
      if (nms_enabled)
        keys[pos] = ((unsigned long long)(~ordered_bits(score)) << 32) | (unsigned)pos;
    }
    V += total;
  }
  for (size_t r = (size_t)V * 7 + tid; r < (size_t)A * 7; r += kTB) po[r] = -1.f;
  const bool do_nms = nms_enabled && V >= 1;
  if (tid == 0) ws.nms_count[b] = do_nms ? V : 0;
  if (!do_nms) return;

  // (a) order by score, descending; keys are unique (score, position) pairs, so the order is that of a stable sort.
  // Only the first nkeep = min(V, nms_topk) sorted rows are ever written back (multibox_detection.cc:143-151).  Round 5: when
  // nms_topk cuts the list, those rows are SELECTED first -- a radix select of the nkeep-th smallest score key over four 8-bit
  // digits, ties at the cut taken in position order -- and only they are sorted (512 keys by one wave instead of 8192 by the
  // workgroup: 91 passes over 64 KiB of LDS were 0.05 ms of this kernel).
  int nkeep = V;
  if (nms_topk > 0 && nms_topk < nkeep) nkeep = nms_topk;
  int n2 = 1;
  while (n2 < V) n2 <<= 1;
  int n2k = 1;
  while (n2k < nkeep) n2k <<= 1;
  const bool select = kLdsKeys && nkeep < V && n2k <= selcap;        // (block-uniform)
  unsigned long long *sorted = keys;
  __syncthreads();
  if (select) {
    unsigned long long *sel = keys + n2cap;
    unsigned prefix = 0;
    int kk = nkeep;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int i = tid; i < 256; i += kTB) s_hist[i] = 0;
      __syncthreads();
      for (int i = tid; i < V; i += kTB) {
        const unsigned hi = (unsigned)(keys[i] >> 32);
        const bool match = (shift == 24) ? true : ((hi >> (shift + 8)) == (prefix >> (shift + 8)));
        if (match) atomicAdd(&s_hist[(hi >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid < 64) {      // first bin whose running count reaches kk: four bins per lane, a shuffle scan over the lanes
        unsigned h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = s_hist[4 * tid + q];
        const int mine = (int)(h[0] + h[1] + h[2] + h[3]);
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (tid >= d) incl += t; }
        const unsigned long long reach = __ballot(incl >= kk);
        const int first = __ffsll((long long)reach) - 1;
        if (tid == first) {
          int cum = incl - mine, bin = 4 * tid;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (cum + (int)h[q] >= kk) { bin = 4 * tid + q; break; }
            cum += (int)h[q];
          }
          s_sel[0] = bin; s_sel[1] = kk - cum; s_sel[2] = (int)s_hist[bin];
        }
      }
      __syncthreads();
      prefix |= (unsigned)s_sel[0] << shift;
      kk = s_sel[1];
    }
    const unsigned T = prefix;
    const int nless = nkeep - kk, ties = s_sel[2];
    if (tid == 0) s_sel[3] = 0;
    __syncthreads();
    if (kk == ties) {          // every row at the cut is kept: any slot will do, the sort below orders them
      for (int i = tid; i < V; i += kTB) {
        const unsigned long long k = keys[i];
        if ((unsigned)(k >> 32) <= T) sel[atomicAdd(&s_sel[3], 1)] = k;
      }
    } else {                   // equal scores across the cut: the first kk of them in position order
      int taken = 0;
      for (int base = 0; base < V; base += kTB) {
        const int i = base + tid;
        const unsigned long long k = i < V ? keys[i] : ~0ull;
        const unsigned hi = (unsigned)(k >> 32);
        const bool tie = i < V && hi == T;
        int total;
        const int off = block_scan_pred(tie, s_w, total);
        if (i < V && hi < T) sel[atomicAdd(&s_sel[3], 1)] = k;
        if (tie && taken + off < kk) sel[nless + taken + off] = k;
        taken += total;
      }
    }
    for (int i = nkeep + tid; i < n2k; i += kTB) sel[i] = ~0ull;
    __syncthreads();
    bitonic_sort_keys<true>(sel, n2k, tid);
    sorted = sel;
  } else {
    for (int i = V + tid; i < n2; i += kTB) keys[i] = ~0ull;
    __syncthreads();
    bitonic_sort_keys<kLdsKeys>(keys, n2, tid);
  }
  for (int e = tid; e < nkeep * 7; e += kTB) {
    const int i = e / 7, q = e - i * 7;
    const unsigned src = (unsigned)(sorted[i] & 0xffffffffull);
    po[e] = pt[(size_t)src * 7 + q];
  }
  // (b) Suppression only ever relates rows of ONE class (multibox_detection.cc:159; every pair under force_suppress), and a
  // row's fate depends only on the earlier rows of its class: the greedy pass over all rows is C - 1 independent greedy
  // passes.  perm = the valid rows ordered by (class id, row): the suppression matrix and the scan work on these GROUPED
  // positions, where a class is a contiguous segment (A^2 / 2 -> sum_c n_c^2 / 2 box pairs, and a chain of n_c / 64 dependent
  // blocks per scan instead of A / 64).  gbox = the boxes in that order with their areas, so that the mask kernel reads
  // contiguous 32-byte records instead of gathering through perm.
  int *perm = ws.perm + (size_t)b * A, *gcls = ws.gcls + (size_t)b * A;
  float4 *gbox = ws.gbox + (size_t)b * A * 2;
  auto put_box = [&](int pos, const float *r, int c) {
    const float x1 = r[2], y1 = r[3], x2 = r[4], y2 = r[5];
    gbox[2 * pos] = make_float4(x1, y1, x2, y2);
    gbox[2 * pos + 1] = make_float4((x2 - x1) * (y2 - y1), (float)c, 0.f, 0.f);
  };
  // the source row of output row i (rows past nms_topk keep their pre-sort contents)
  auto source = [&](int i) { return i < nkeep ? (unsigned)(sorted[i] & 0xffffffffull) : (unsigned)i; };
  if (force) {
    for (int i = tid; i < V; i += kTB) { perm[i] = i; gcls[i] = 0; put_box(i, pt + (size_t)source(i) * 7, 0); }
    return;
  }
  const int ncls = C - 1;
  if (ncls <= kCountCls) {
    // Round 5: a stable counting sort by class instead of a second 8192-key bitonic sort.  Wave w owns the contiguous rows
    // [w R, (w+1) R); inside a 64-row step the rows of one class are found with a ballot per class present.
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < (kTB / 64) * kCountCls; i += kTB) (&s_ccnt[0][0])[i] = 0;
    __syncthreads();
    const int R = ((V + kTB - 1) / kTB) * 64;
    const int r0 = wave * R, r1 = min(V, r0 + R);
    for (int base = r0; base < r1; base += 64) {
      const int i = base + lane;
      const bool valid = i < r1;
      const int c = valid ? (int)pt[(size_t)source(i) * 7] : -1;
      unsigned long long todo = __ballot(valid);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int cc = __shfl(c, leader, 64);
        const unsigned long long m = __ballot(valid && c == cc);
        if (lane == leader) atomicAdd(&s_ccnt[wave][cc], (int)__popcll(m));
        todo &= ~m;
      }
    }
    __syncthreads();
    for (int c = tid; c < ncls; c += kTB) {       // exclusive prefix over the waves, per class
      int run = 0;
      for (int w = 0; w < kTB / 64; ++w) { const int t = s_ccnt[w][c]; s_ccnt[w][c] = run; run += t; }
      s_ctot[c] = run;
    }
    __syncthreads();
    if (tid < 64) {                               // exclusive prefix of the class totals
      int t[4], mine = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int idx = 4 * tid + q; t[q] = idx < ncls ? s_ctot[idx] : 0; mine += t[q]; }
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (tid >= d) incl += v; }
      int excl = incl - mine;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int idx = 4 * tid + q; if (idx < ncls) s_ctot[idx] = excl; excl += t[q]; }
    }
    __syncthreads();
    for (int base = r0; base < r1; base += 64) {
      const int i = base + lane;
      const bool valid = i < r1;
      const unsigned src = valid ? source(i) : 0u;
      const int c = valid ? (int)pt[(size_t)src * 7] : -1;
      unsigned long long todo = __ballot(valid);
      int pos = 0;
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int cc = __shfl(c, leader, 64);
        const unsigned long long m = __ballot(valid && c == cc);
        int start = 0;
        if (lane == leader) start = s_ctot[cc] + atomicAdd(&s_ccnt[wave][cc], (int)__popcll(m));
        start = __shfl(start, leader, 64);
        if (valid && c == cc) pos = start + (int)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
      }
      if (valid) { perm[pos] = i; gcls[pos] = c; put_box(pos, pt + (size_t)src * 7, c); }
    }
    return;
  }
  // more classes than the counting table holds: a second sort, of (class, row) keys
  for (int i = tid; i < n2; i += kTB) {     // (entry i is read and rewritten by the same thread)
    unsigned long long k2 = ~0ull;
    if (i < V) k2 = ((unsigned long long)(unsigned)(int)pt[(size_t)source(i) * 7] << 32) | (unsigned)i;
    keys[i] = k2;
  }
  __syncthreads();
  bitonic_sort_keys<kLdsKeys>(keys, n2, tid);
  for (int i = tid; i < V; i += kTB) {
    const int r = (int)(unsigned)(keys[i] & 0xffffffffull), c = (int)(keys[i] >> 32);
    perm[i] = r;
    gcls[i] = c;
    put_box(i, po + (size_t)r * 7, c);      // (output rows: written above by this workgroup, several barriers ago)
  }
}

__device__ __forceinline__ float nms_iou(const float *a, const float *b) {
  const float w = fmaxr(0.f, fminr(a[2], b[2]) - fmaxr(a[0], b[0]));
  const float h = fmaxr(0.f, fminr(a[3], b[3]) - fmaxr(a[1], b[1]));
  const float i = w * h;
  const float u = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - i;
  return u <= 0.f ? 0.f : i / u;
}

// 64x64 tiles of the upper-triangular suppression matrix over the GROUPED positions (ws.perm: rows ordered by class, then
// row): bit (i,j), j>i: grouped row i would suppress grouped row j (multibox_detection.cc:153-167).
// One four-wave workgroup per (64-row tile, sample); wave q takes the column tiles rt + q, rt + q + 4, ...  A row tile only
// meets the column tiles up to the end of its last row's class segment; the others are never read by the scan.
// The kernel runs on MultiBoxDetection's side stream beside the training step, and the dispatcher hands out the workgroups
// of ONE queue's kernel at a time: with 295 000 tiny workgroups (96 x 96 tiles x 32 samples, the first version) every
// main-stream kernel that became ready meanwhile waited for the whole dispatch (a 7-us BatchNorm table merge took 0.46 ms,
// rocprofv3 kernel trace of the step); a few thousand longer-lived ones are handed out in microseconds.
// Round 5 (0.24 -> see DESIGN.md): a lane owns a COLUMN; the 64 rows of the tile pass by as wave-uniform values (v_readlane of
// the 32-byte grouped records det_decode_sort_kernel left: box, area, class) -- no LDS, no barrier, no gather through perm --
// and row t's 64-bit word is the ballot of its comparison, written into lane t's registers.  The division of IoU >= threshold is only executed when
// some lane's products are within 2^-20 of the cut: i >= t_hi u (t_hi = threshold (1 + 2^-20)) implies the rounded quotient
// reaches the threshold, i <= t_lo u implies it does not (rounding is monotone and the threshold is a float), and boxes that
// do not overlap at all (i = 0), nearly all of them, are in the second class.  A tile holding a non-finite (or absurdly large)
// coordinate takes the literal expression (the reference's min / max are not IEEE minNum / maxNum on NaN).
constexpr int kMaskWaves = 4;
__global__ __launch_bounds__(64 * kMaskWaves) void nms_mask_kernel(const float4 *__restrict__ gbox_all, const int *__restrict__ gcls_all,
                                                      const int *__restrict__ nms_count,
                                                      unsigned long long *__restrict__ mask_all, int A,
                                                      int nwords, float nms_threshold, int force) {
  const int rt = blockIdx.x, b = blockIdx.y;
  const int V = nms_count[b];
  const int ntile = (V + 63) >> 6;
  const int ct0 = rt + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (the wave index, as a scalar)
  if (rt * 64 >= V || ct0 >= ntile) return;
  const float4 *gbox = gbox_all + (size_t)b * A * 2;
  const int *gcls = gcls_all + (size_t)b * A;
  const int lane = threadIdx.x & 63;
  const int nrow = min(64, V - rt * 64);
  // this lane's row of the tile; row t reaches the wave as scalars through v_readlane (no memory latency inside the loop)
  float4 rb = make_float4(0.f, 0.f, 0.f, 0.f), re = make_float4(0.f, -1.f, 0.f, 0.f);
  if (lane < nrow) { rb = gbox[2 * (rt * 64 + lane)]; re = gbox[2 * (rt * 64 + lane) + 1]; }
  // "tame": every coordinate is a number below 1e15 in magnitude -- no product below can overflow or be NaN
  auto tame4 = [](const float4 v) { return fabsf(v.x) <= 1e15f && fabsf(v.y) <= 1e15f && fabsf(v.z) <= 1e15f && fabsf(v.w) <= 1e15f; };
  const bool rows_tame = __ballot(!tame4(rb)) == 0ull;
  const float t_hi = nms_threshold * (1.0f + 0x1p-20f), t_lo = nms_threshold * (1.0f - 0x1p-20f);
  // class of this tile's first and last row: grouped positions ascend in class, so a column tile that starts above the last
  // row's class has no pair, and one that ends in the first row's class holds that class only
  const int c_lo = gcls[rt * 64], c_hi = gcls[rt * 64 + nrow - 1];
  unsigned long long *mrow = mask_all + ((size_t)b * A + rt * 64 + lane) * nwords;
  // lane t of (lo, hi) := a wave-uniform 64-bit word.  (On gfx9 the lane select of v_writelane beside a scalar source has to
  // be M0; nothing else in this kernel uses M0 -- the compiler reserves it and warns about the clobber.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  auto put_word = [](int &lo, int &hi, unsigned long long word, int t) {
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                 : "+v"(lo), "+v"(hi) : "s"((int)(unsigned)word), "s"((int)(unsigned)(word >> 32)), "s"(t) : "m0");
  };
#pragma clang diagnostic pop
  auto vmin_sv = [](float sv, float vv) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(sv), "v"(vv)); return r; };
  auto vmax_sv = [](float sv, float vv) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(sv), "v"(vv)); return r; };
  auto rl = [](float v, int t) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t)); };
  // a column tile: this lane's record and the classes of the tile's first and last row; the NEXT tile of the wave is
  // requested before the current one is worked on (its latency, ~2 us, would otherwise stand between two 3-us loops)
  struct Col { float4 cb, ce; int c_first, c_last; };
  auto fetch = [&](int ct) {
    Col c;
    c.cb = make_float4(0.f, 0.f, 0.f, 0.f); c.ce = make_float4(0.f, -2.f, 0.f, 0.f); c.c_first = 0x7fffffff; c.c_last = 0;
    if (ct < ntile) {
      const int jc = ct * 64 + lane;
      if (jc < V) { c.cb = gbox[2 * jc]; c.ce = gbox[2 * jc + 1]; }
      c.c_first = gcls[ct * 64];
      c.c_last = gcls[min(V, ct * 64 + 64) - 1];
    }
    return c;
  };
  Col nxt = fetch(ct0);
  for (int ct = ct0; ct < ntile; ct += kMaskWaves) {
    const Col cur = nxt;
    if (cur.c_first > c_hi) break;        // (wave-uniform) this and every later column tile hold other classes only
    nxt = fetch(ct + kMaskWaves);
    const bool colv = ct * 64 + lane < V;
    const float4 cb = cur.cb, ce = cur.ce;
    const bool diag = ct == rt;
    // every pair of the two tiles is of one class (or classes do not matter), and no pair is below the diagonal
    const bool plain = !diag && (force || (c_lo == c_hi && cur.c_last == c_lo));
    int bits_lo = 0, bits_hi = 0;
    const unsigned long long colv_m = __ballot(colv);
    if (rows_tame && __ballot(!tame4(cb)) == 0ull) {
      for (int t = 0; t < nrow; ++t) {
        const float rx = rl(rb.x, t), ry = rl(rb.y, t), rz = rl(rb.z, t), rw = rl(rb.w, t), ra = rl(re.x, t);
        // (tame numbers: the bare v_min / v_max instructions, without the quieting copies fminf / fmaxf come with)
        const float w = fmaxf(0.f, vmin_sv(rz, cb.z) - vmax_sv(rx, cb.x));
        const float h = fmaxf(0.f, vmin_sv(rw, cb.w) - vmax_sv(ry, cb.y));
        const float i = w * h;
        const float u = ra + ce.x - i;
        unsigned long long elig = colv_m;
        if (!plain) {
          if (diag) elig &= (~0ull << t) << 1;               // (columns right of the diagonal: lanes > t)
          if (!force) elig &= __ballot(rl(re.y, t) == ce.y);
        }
        // certain without dividing: yes if i >= t_hi u, no if i <= t_lo u (u in the normal range), no if u <= 0
        const unsigned long long normal = __ballot(u >= 1e-30f), ge_hi = __ballot(i >= t_hi * u);
        const unsigned long long le_lo = __ballot(i <= t_lo * u), pos = __ballot(u > 0.f);
        unsigned long long yes = normal & ge_hi;
        const unsigned long long sure = (normal & (ge_hi | le_lo)) | ~pos;
        if (elig & ~sure) yes = __ballot(u > 0.f && i / u >= nms_threshold);
        put_word(bits_lo, bits_hi, elig & yes, t);
      }
    } else {
      for (int t = 0; t < nrow; ++t) {
        float a[4], c[4];
        a[0] = rl(rb.x, t); a[1] = rl(rb.y, t); a[2] = rl(rb.z, t); a[3] = rl(rb.w, t);
        c[0] = cb.x; c[1] = cb.y; c[2] = cb.z; c[3] = cb.w;
        const bool elig = colv && (force || rl(re.y, t) == ce.y) && (!diag || lane > t);
        const unsigned long long word = __ballot(elig && nms_iou(a, c) >= nms_threshold);
        put_word(bits_lo, bits_hi, word, t);
      }
    }
    const unsigned long long bits = ((unsigned long long)(unsigned)bits_hi << 32) | (unsigned)bits_lo;
    if (lane < nrow) mrow[ct] = bits;
  }
}

// Greedy scan over the suppression bit matrix, one 16-wave workgroup per (sample, class): the grouped positions [s, e) of
// class blockIdx.y (every valid row under force_suppress).  Rows are resolved in blocks of 64 grouped positions: wave 0
// walks the block's 64x64 diagonal word by word with lane shuffles (the only inherently serial part); the surviving rows
// then OR their mask words into the `removed` words of all later blocks of the segment -- 64 rows x up to 128 words, one
// (row, word phase) pair per thread, LDS atomics (OR is order independent, so the result is deterministic).
// The mask words a block needs do not depend on which rows survive, so they are requested one block AHEAD and are in
// registers by the time the diagonal is resolved: per block the critical path is the 64-step shuffle chain plus two
// barriers instead of a chain of dependent global loads (the one-wave version took 1.4 ms for 6132 rows).
// The first and last block of a segment also hold rows of the neighbouring classes: no mask bit relates them to this
// class's rows, so whatever this workgroup concludes about them is ignored (only rows of [s, e) are written).
constexpr int kScanThreads = 1024, kScanPhases = kScanThreads / 64, kScanPre = 8;   // 8 x 16 = 128 words ahead per row
__global__ __launch_bounds__(kScanThreads) void nms_scan_kernel(float *__restrict__ out, int A, int nwords, int force,
                                                                DetWs ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *removed = reinterpret_cast<unsigned long long *>(smem);
  __shared__ unsigned long long s_alive;
  __shared__ int s_seg[2];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = ws.nms_count[b];
  if (V == 0) return;
  float *po = out + (size_t)b * A * 7;
  const int *perm = ws.perm + (size_t)b * A;
  if (tid == 0) {
    int s0 = 0, e0 = V;
    if (!force) {      // lower bounds of class c and c + 1 among the grouped positions (ascending in class)
      const int *gcls = ws.gcls + (size_t)b * A;
      const int c = (int)blockIdx.y;
      int lo = 0, hi = V;
      while (lo < hi) { const int m = (lo + hi) >> 1; if (gcls[m] < c) lo = m + 1; else hi = m; }
      s0 = lo; hi = V;
      while (lo < hi) { const int m = (lo + hi) >> 1; if (gcls[m] <= c) lo = m + 1; else hi = m; }
      e0 = lo;
    }
    s_seg[0] = s0; s_seg[1] = e0;
  }
  __syncthreads();
  const int seg_s = s_seg[0], seg_e = s_seg[1];
  if (seg_e <= seg_s) return;
  const int wb = seg_s >> 6;                    // first block of the segment
  const int nw = ((seg_e - 1) >> 6) + 1;        // one past its last block
  const unsigned long long *mask = ws.mask + (size_t)b * A * nwords;
  for (int w = wb + tid; w < nw; w += kScanThreads) removed[w] = 0;
  const int r = tid / kScanPhases, ph = tid % kScanPhases;     // propagation role: row r of the block, words ph, ph+16, ...
  unsigned long long v[kScanPre], vn[kScanPre];
  auto preload = [&](int w0, unsigned long long *dst) {
    const int row = w0 * 64 + r;
#pragma unroll
    for (int k = 0; k < kScanPre; ++k) {
      const int w = w0 + 1 + ph + kScanPhases * k;
      dst[k] = (w < nw && row >= seg_s && row < seg_e) ? mask[(size_t)row * nwords + w] : 0ull;
    }
  };
  auto load_diag = [&](int w0) {
    const int row = w0 * 64 + lane;
    return (wave == 0 && row >= seg_s && row < seg_e) ? mask[(size_t)row * nwords + w0] : 0ull;
  };
  preload(wb, v);
  unsigned long long diag = load_diag(wb), diag_n = 0;
  __syncthreads();
  for (int w0 = wb; w0 < nw; ++w0) {
    if (w0 + 1 < nw) { preload(w0 + 1, vn); diag_n = load_diag(w0 + 1); }   // in flight while this block is resolved
    if (wave == 0) {
      // (round 5: the 64-step chain in scalar registers -- v_readlane of row t's word, not a trip through the LDS crossbar
      // per step: 3 us -> 0.3 us per block, and a class of 766 rows is a chain of 12 blocks)
      const unsigned long long cur0 = removed[w0];
      unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur0 >> 32)) << 32) |
                               (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur0);
      const int dlo = (int)(unsigned)diag, dhi = (int)(unsigned)(diag >> 32);
      for (int t = 0; t < 64; ++t) {
        const unsigned long long d = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, t) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane(dlo, t);
        if (!((cur >> t) & 1ull)) cur |= d;
      }
      const int lo = max(seg_s - w0 * 64, 0), hi = min(seg_e - w0 * 64, 64);     // this segment's rows of the block
      const unsigned long long below = hi == 64 ? ~0ull : ((1ull << hi) - 1ull);
      const unsigned long long valid = below & ~((1ull << lo) - 1ull);
      if (lane == 0) { removed[w0] = cur; s_alive = ~cur & valid; }
    }
    __syncthreads();
    if ((s_alive >> r) & 1ull) {
#pragma unroll
      for (int k = 0; k < kScanPre; ++k)
        if (v[k]) atomicOr(&removed[w0 + 1 + ph + kScanPhases * k], v[k]);
      // more than 128 later words (A > 8256 anchors): the rest is fetched now
      const int row = w0 * 64 + r;
      for (int w = w0 + 1 + ph + kScanPhases * kScanPre; w < nw; w += kScanPhases) {
        const unsigned long long m = mask[(size_t)row * nwords + w];
        if (m) atomicOr(&removed[w], m);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kScanPre; ++k) v[k] = vn[k];
    diag = diag_n;
  }
  for (int i = seg_s + tid; i < seg_e; i += kScanThreads)
    if ((removed[i >> 6] >> (i & 63)) & 1ull) po[(size_t)perm[i] * 7] = -1.f;
}

struct TargetLayout { size_t err, ngt, row_iou, row_gt, bgkey, flag, cpart, total; int nblk; };
TargetLayout target_layout(int B, int A, int L) {
  TargetLayout l;
  size_t o = 0;
  l.err = o; o = dspn::align_up(o + sizeof(int) * B, 256);
  l.ngt = o; o = dspn::align_up(o + sizeof(int) * B, 256);
  l.row_iou = o; o = dspn::align_up(o + sizeof(float) * (size_t)B * A, 256);
  l.row_gt = o; o = dspn::align_up(o + sizeof(int) * (size_t)B * A, 256);
  l.bgkey = o; o = dspn::align_up(o + sizeof(unsigned) * (size_t)B * A, 256);
  l.flag = o; o = dspn::align_up(o + (size_t)B * A, 256);
  l.nblk = (A + 255) / 256;
  l.cpart = o; o = dspn::align_up(o + 8 * (size_t)B * l.nblk * (L > 0 ? L : 1), 256);
  l.total = o;
  return l;
}

constexpr int kLdsKeyCap = 16384;  // 128 KiB of 64-bit keys
int next_pow2(int v) { int n = 1; while (n < v) n <<= 1; return n; }
struct DetLayout { size_t cnt, temp, keys, mask, perm, gcls, gbox, total; int n2cap, nwords; bool lds_keys; };
DetLayout det_layout(int B, int A) {
  DetLayout l;
  l.n2cap = next_pow2(A);
  l.lds_keys = l.n2cap <= kLdsKeyCap;
  l.nwords = (A + 63) / 64;
  size_t o = 0;
  l.cnt = o; o = dspn::align_up(o + sizeof(int) * B, 256);
  l.temp = o; o = dspn::align_up(o + sizeof(float) * (size_t)B * A * 7, 256);
  l.keys = o; if (!l.lds_keys) o = dspn::align_up(o + 8 * (size_t)B * l.n2cap, 256);
  l.mask = o; o = dspn::align_up(o + 8 * (size_t)B * A * l.nwords, 256);
  l.perm = o; o = dspn::align_up(o + sizeof(int) * (size_t)B * A, 256);
  l.gcls = o; o = dspn::align_up(o + sizeof(int) * (size_t)B * A, 256);
  l.gbox = o; o = dspn::align_up(o + 32 * (size_t)B * A, 256);
  l.total = o;
  return l;
}

}  // namespace

extern "C" {

int dspn_multibox_prior_f32(const float *sizes, int num_sizes, const float *ratios, int num_ratios,
                            int in_height, int in_width, float step_y, float step_x,
                            float offset_y, float offset_x, int clip, float *out_dev,
                            void *stream) {
  // attribute checks of MultiBoxPriorOp's constructor (multibox_prior-inl.h:88-96)
  DSPN_REQUIRE(sizes && num_sizes > 0, "MultiBoxPrior: sizes must not be empty");
  DSPN_REQUIRE(ratios && num_ratios > 0, "MultiBoxPrior: ratios must not be empty");
  DSPN_REQUIRE(num_sizes <= kMaxAttr && num_ratios <= kMaxAttr,
               "MultiBoxPrior: at most %d sizes / ratios", kMaxAttr);
  DSPN_REQUIRE(in_height > 0, "Input height should > 0");
  DSPN_REQUIRE(in_width > 0, "Input width should > 0");
  DSPN_REQUIRE(offset_y >= 0.f && offset_y <= 1.f && offset_x >= 0.f && offset_x <= 1.f,
               "MultiBoxPrior: offsets must be in [0,1]");
  DSPN_REQUIRE(step_y * step_x >= 0, "Must specify both step_y and step_x");
  DSPN_REQUIRE(out_dev, "MultiBoxPrior: null output");
  if (step_y <= 0 || step_x <= 0) { step_y = 1.f / in_height; step_x = 1.f / in_width; }
  PriorAttr attr;
  memset(&attr, 0, sizeof(attr));
  memcpy(attr.sizes, sizes, sizeof(float) * num_sizes);
  memcpy(attr.ratios, ratios, sizeof(float) * num_ratios);
  const int total = in_height * in_width * (num_sizes + num_ratios - 1);
  hipLaunchKernelGGL(prior_kernel, dim3(dspn::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     attr, num_sizes, num_ratios, in_height, in_width, step_y, step_x, offset_y,
                     offset_x, clip, reinterpret_cast<float4 *>(out_dev));
  return dspn::check_launch("multibox_prior");
}

size_t dspn_multibox_target_workspace_bytes(int batch, int num_anchors, int num_labels) {
  if (batch <= 0 || num_anchors <= 0) return 0;
  return target_layout(batch, num_anchors, num_labels).total;
}

int dspn_multibox_target_f32(const float *anchors_dev, const float *labels_dev,
                             const float *cls_preds_dev, int batch, int num_anchors,
                             int num_labels, int label_width, int num_classes,
                             float overlap_threshold, float ignore_label,
                             float negative_mining_ratio, float negative_mining_thresh,
                             int minimum_negative_samples, const float variances[4],
                             float *loc_target_dev, float *loc_mask_dev, float *cls_target_dev,
                             void *workspace_dev, size_t workspace_bytes, void *stream) {
  (void)minimum_negative_samples;  // CPU reference ignores it (multibox_target.cc:186-189)
  // shape rules of MultiBoxTargetProp::InferShape (multibox_target-inl.h:213-238)
  DSPN_REQUIRE(batch > 0, "MultiBoxTarget: batch must be > 0");
  DSPN_REQUIRE(num_anchors > 0, "Number boxes should > 0");
  DSPN_REQUIRE(num_labels > 0, "Padded label should > 0");
  DSPN_REQUIRE(label_width == 6, "Label width should be 6: [cls-xmin-ymin-xmax-ymax-dist]");
  DSPN_REQUIRE(num_classes > 0, "Prediction: [nbatch-num_classes-num_anchors]");
  DSPN_REQUIRE(num_labels <= kMaxLabels, "MultiBoxTarget: at most %d label rows", kMaxLabels);
  DSPN_REQUIRE(variances, "MultiBoxTarget: variances must have 4 values");
  if (negative_mining_ratio > 0)
    DSPN_REQUIRE(negative_mining_thresh > 0, "negative_mining_thresh must be > 0");
  DSPN_REQUIRE(anchors_dev && labels_dev && cls_preds_dev && loc_target_dev && loc_mask_dev &&
                   cls_target_dev && workspace_dev, "MultiBoxTarget: null pointer");
  const TargetLayout l = target_layout(batch, num_anchors, num_labels);
  if (workspace_bytes < l.total)
    return dspn::fail(DSPN_ERR_WORKSPACE_, "MultiBoxTarget: workspace %zu < %zu bytes",
                      workspace_bytes, l.total);
  char *w = static_cast<char *>(workspace_dev);
  TargetWs ws;
  ws.err = reinterpret_cast<int *>(w + l.err);
  ws.ngt = reinterpret_cast<int *>(w + l.ngt);
  ws.row_iou = reinterpret_cast<float *>(w + l.row_iou);
  ws.row_gt = reinterpret_cast<int *>(w + l.row_gt);
  ws.bgkey = reinterpret_cast<unsigned *>(w + l.bgkey);
  ws.flag = reinterpret_cast<signed char *>(w + l.flag);
  ws.cpart = reinterpret_cast<unsigned long long *>(w + l.cpart);
  ws.nblk = l.nblk;
  hipStream_t s = (hipStream_t)stream;
  const float4 *an = reinterpret_cast<const float4 *>(anchors_dev);
  const dim3 grid(dspn::cdiv(num_anchors, 256), batch);
  hipLaunchKernelGGL(target_rows_kernel, grid, dim3(256), 0, s, an, labels_dev, cls_preds_dev,
                     num_anchors, num_labels, label_width, num_classes, ws);
  if (num_anchors <= 8 * kTB)
    hipLaunchKernelGGL(target_match_reg_kernel<8>, dim3(batch), dim3(kTB), 0, s, an, labels_dev,
                       num_anchors, num_labels, label_width, overlap_threshold,
                       negative_mining_ratio, negative_mining_thresh, ws);
  else
    hipLaunchKernelGGL(target_match_kernel, dim3(batch), dim3(kTB), 0, s, an, labels_dev,
                       num_anchors, num_labels, label_width, overlap_threshold,
                       negative_mining_ratio, negative_mining_thresh, ws);
  hipLaunchKernelGGL(target_write_kernel, grid, dim3(256), 0, s, an, labels_dev, num_anchors,
                     num_labels, label_width, ignore_label, variances[0], variances[1],
                     variances[2], variances[3], ws, loc_target_dev, loc_mask_dev, cls_target_dev);
  return dspn::check_launch("multibox_target");
}

int dspn_multibox_target_errors(const void *workspace_dev, int batch, int *host_codes,
                                void *stream) {
  DSPN_REQUIRE(workspace_dev && host_codes && batch > 0, "multibox_target_errors: bad argument");
  hipError_t e = hipMemcpyAsync(host_codes, workspace_dev, sizeof(int) * batch,
                                hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess)
    return dspn::fail(DSPN_ERR_LAUNCH_, "multibox_target_errors: %s", hipGetErrorString(e));
  for (int i = 0; i < batch; ++i)
    if (host_codes[i] != 0) {
      dspn::fail(-host_codes[i], host_codes[i] == 2
                     ? "MultiBoxTarget: sample %d: padded label row is not all -1"
                     : "MultiBoxTarget: sample %d: fewer mining candidates than negatives", i);
      return -host_codes[i];
    }
  return 0;
}

size_t dspn_multibox_detection_workspace_bytes(int batch, int num_anchors) {
  if (batch <= 0 || num_anchors <= 0) return 0;
  return det_layout(batch, num_anchors).total;
}

int dspn_multibox_detection_f32(const float *cls_prob_dev, const float *loc_pred_dev,
                                const float *anchors_dev, int batch, int num_anchors,
                                int num_classes, float threshold, int clip,
                                const float variances[4], float nms_threshold,
                                int force_suppress, int nms_topk, float *out_dev,
                                void *workspace_dev, size_t workspace_bytes, void *stream) {
  // shape rules of MultiBoxDetectionProp::InferShape (multibox_detection-inl.h:149-171)
  DSPN_REQUIRE(batch > 0, "MultiBoxDetection: batch must be > 0");
  DSPN_REQUIRE(num_anchors > 0, "Number of anchors must > 0");
  DSPN_REQUIRE(num_classes > 0, "MultiBoxDetection: num_classes must be > 0");
  DSPN_REQUIRE(variances, "Variance size must be 4");
  DSPN_REQUIRE(cls_prob_dev && loc_pred_dev && anchors_dev && out_dev && workspace_dev,
               "MultiBoxDetection: null pointer");
  const DetLayout l = det_layout(batch, num_anchors);
  if (workspace_bytes < l.total)
    return dspn::fail(DSPN_ERR_WORKSPACE_, "MultiBoxDetection: workspace %zu < %zu bytes",
                      workspace_bytes, l.total);
  char *w = static_cast<char *>(workspace_dev);
  DetWs ws;
  ws.nms_count = reinterpret_cast<int *>(w + l.cnt);
  ws.temp = reinterpret_cast<float *>(w + l.temp);
  ws.keys = reinterpret_cast<unsigned long long *>(w + l.keys);
  ws.mask = reinterpret_cast<unsigned long long *>(w + l.mask);
  ws.perm = reinterpret_cast<int *>(w + l.perm);
  ws.gcls = reinterpret_cast<int *>(w + l.gcls);
  ws.gbox = reinterpret_cast<float4 *>(w + l.gbox);
  hipStream_t s = (hipStream_t)stream;
  const float4 *an = reinterpret_cast<const float4 *>(anchors_dev);
  const int nms_enabled = !(nms_threshold <= 0 || nms_threshold > 1);
  if (l.lds_keys) {
    // keys of the whole list + (when nms_topk cuts it to at most half) the selected keys, beside ~18 KiB of static LDS
    constexpr size_t kDynMax = 140 * 1024;
    int selcap = nms_topk > 0 ? next_pow2(std::min(nms_topk, num_anchors)) : 0;
    if (!nms_enabled || 2 * selcap > l.n2cap || 8 * (size_t)(l.n2cap + selcap) > kDynMax) selcap = 0;
    const size_t lds = 8 * (size_t)(l.n2cap + selcap);
    // (kDynMax + the kernel's ~18.5 KiB of static LDS is ~158 of gfx950's 160 KiB: a part with less, or one more static array,
    // fails in ensure_dynamic_lds with a message instead of at the launch)
    static dspn::KernelDeviceState st;
    if (const int dev = dspn::ensure_dynamic_lds(reinterpret_cast<const void *>(det_decode_sort_kernel<true>), kDynMax, st, "multibox_detection"); dev < 0) return dev;
    hipLaunchKernelGGL(det_decode_sort_kernel<true>, dim3(batch), dim3(kTB), lds, s, cls_prob_dev,
                       loc_pred_dev, an, num_anchors, num_classes, threshold, clip, variances[0],
                       variances[1], variances[2], variances[3], nms_enabled, nms_topk, force_suppress != 0, l.n2cap, selcap,
                       ws, out_dev);
  } else {
    hipLaunchKernelGGL(det_decode_sort_kernel<false>, dim3(batch), dim3(kTB), 0, s, cls_prob_dev,
                       loc_pred_dev, an, num_anchors, num_classes, threshold, clip, variances[0],
                       variances[1], variances[2], variances[3], nms_enabled, nms_topk, force_suppress != 0, l.n2cap, 0,
                       ws, out_dev);
  }
  if (nms_enabled) {
    const int nt = l.nwords;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nt, batch), dim3(64 * kMaskWaves), 0, s, ws.gbox, ws.gcls,
                       ws.nms_count, ws.mask, num_anchors, l.nwords, nms_threshold, force_suppress != 0);
    // one scan per (sample, class id 0 .. num_classes - 2); a single one per sample under force_suppress
    hipLaunchKernelGGL(nms_scan_kernel, dim3(batch, (force_suppress || num_classes < 3) ? 1 : num_classes - 1), dim3(kScanThreads),
                       8 * (size_t)l.nwords, s, out_dev, num_anchors, l.nwords, force_suppress != 0, ws);
  }
  return dspn::check_launch("multibox_detection");
}

}  // extern "C"
