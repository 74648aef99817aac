// Pixel-coordinate ("+1" convention) greedy NMS for MI355X (gfx950); C ABI in include/dspn_nms.h.
//
// Three kernels on the caller's stream: (1) one workgroup sorts (score, index) keys in LDS (bitonic, total order:
// score descending, then index descending); (2) a 64 x 64-tile suppression bit matrix over the sorted order, one
// wave64 per tile: lane r owns row r of the tile and tests it against the 64 column boxes held in LDS -- the
// reference's CUDA kernel has the same shape with 64 threads because its mask word is 64 bits (nms_kernel.cu:21,
// 34-78), which is exactly one CDNA wavefront; (3) one wave walks the row blocks in order, resolving the 64 rows of
// a block with wave shuffles and OR-ing the masks of the surviving rows into the running `removed` words -- the
// reference copies the whole matrix to the host and does this loop on the CPU (nms_kernel.cu:124-139).
// Float arithmetic follows the reference's operation order; this file is compiled with -ffp-contract=off.
#include "dspn_common.h"
#include "../../include/dspn_nms.h"

namespace {

constexpr int kMaxN = 8192;

__device__ __forceinline__ bool key_before(float sa, int ia, float sb, int ib) {
  // total order of the sorted sequence: higher score first, then higher index first
  return sa > sb || (sa == sb && ia > ib);
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(const float *__restrict__ dets, int n, int npow2,
                                                        int *__restrict__ order, float *__restrict__ sorted) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  float *ks = reinterpret_cast<float *>(smraw);          // [npow2] scores
  int *ki = reinterpret_cast<int *>(ks + npow2);         // [npow2] indices
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    ks[i] = i < n ? dets[i * 5 + 4] : -INFINITY;
    ki[i] = i < n ? i : -1 - i;                          // padding sorts last (and NaN scores are the caller's problem)
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const float sa = ks[i], sb = ks[l];
          const int ia = ki[i], ib = ki[l];
          const bool swap = up ? key_before(sb, ib, sa, ia) : key_before(sa, ia, sb, ib);
          if (swap) { ks[i] = sb; ks[l] = sa; ki[i] = ib; ki[l] = ia; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int src = ki[i];
    order[i] = src;
#pragma unroll
    for (int e = 0; e < 4; ++e) sorted[i * 4 + e] = dets[src * 5 + e];
  }
}

__global__ __launch_bounds__(64) void nms_mask_kernel_px(const float *__restrict__ boxes, int n, float thresh,
                                                         int suppress_ge, unsigned long long *__restrict__ mask,
                                                         int col_blocks) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;                                   // the scan only reads words at or after the row's own block
  __shared__ float cbx[64 * 4];
  const int lane = threadIdx.x;
  const int col = cb * 64 + lane;
  if (col < n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) cbx[lane * 4 + e] = boxes[col * 4 + e];
  }
  __syncthreads();
  const int row = rb * 64 + lane;
  if (row >= n) return;
  const float ax1 = boxes[row * 4], ay1 = boxes[row * 4 + 1], ax2 = boxes[row * 4 + 2], ay2 = boxes[row * 4 + 3];
  const float sa = (ax2 - ax1 + 1.f) * (ay2 - ay1 + 1.f);
  const int ncol = min(64, n - cb * 64);
  unsigned long long t = 0;
  for (int i = (rb == cb ? lane + 1 : 0); i < ncol; ++i) {
    const float bx1 = cbx[i * 4], by1 = cbx[i * 4 + 1], bx2 = cbx[i * 4 + 2], by2 = cbx[i * 4 + 3];
    const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1) + 1.f, 0.f);
    const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1) + 1.f, 0.f);
    const float inter = w * h;
    const float sb = (bx2 - bx1 + 1.f) * (by2 - by1 + 1.f);
    const float ovr = inter / (sa + sb - inter);
    if (suppress_ge == 1 ? ovr >= thresh : (suppress_ge == 2 ? ovr > thresh : !(ovr <= thresh))) t |= 1ull << i;
  }
  mask[(long long)row * col_blocks + cb] = t;
}

__global__ __launch_bounds__(64) void nms_scan_kernel_px(const unsigned long long *__restrict__ mask,
                                                         const int *__restrict__ order, int n, int col_blocks,
                                                         int *__restrict__ keep, int *__restrict__ num_keep) {
  __shared__ unsigned long long removed[kMaxN / 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < col_blocks; i += 64) removed[i] = 0;
  __syncthreads();
  int nkeep = 0;
  for (int b = 0; b < col_blocks; ++b) {
    const int row = b * 64 + lane;
    const unsigned long long own = row < n ? mask[(long long)row * col_blocks + b] : 0ull;   // within-block mask of my row
    unsigned long long rem = removed[b];
    const int rows_here = min(64, n - b * 64);
    unsigned long long kept_bits = 0;
    for (int r = 0; r < rows_here; ++r) {           // resolve the block's rows in order (wave-uniform loop)
      const unsigned long long m = __shfl(own, r);  // mask word of row r
      if (!((rem >> r) & 1ull)) { kept_bits |= 1ull << r; rem |= m; }
    }
    // compact the kept rows of this block into keep[]
    if ((kept_bits >> lane) & 1ull) {
      const int pos = nkeep + __popcll(kept_bits & ((1ull << lane) - 1ull));
      keep[pos] = order[row];
    }
    nkeep += __popcll(kept_bits);
    // the surviving rows suppress later blocks
    for (int j = b + 1 + lane; j < col_blocks; j += 64) {
      unsigned long long acc = removed[j];
      for (int r = 0; r < rows_here; ++r)
        if ((kept_bits >> r) & 1ull) acc |= mask[(long long)(b * 64 + r) * col_blocks + j];
      removed[j] = acc;
    }
    __syncthreads();
  }
  if (lane == 0) *num_keep = nkeep;
}

int pow2_at_least(int n) { int p = 64; while (p < n) p <<= 1; return p; }

// cython/bbox.pyx:15-55 (bbox_overlaps_cython): overlaps[n][k] of (N,4) boxes against (K,4) query boxes, float64, "+1"
// pixel convention, the reference's operation order (this file is built with -ffp-contract=off; the f64 division is
// IEEE).  A workgroup takes 256 consecutive rows n against a block of up to 64 queries held in LDS with their areas;
// each thread owns one row and writes its K doubles in order (the reference's K is a few dozen ground-truth boxes).
__global__ __launch_bounds__(256) void bbox_overlaps_kernel(const double *__restrict__ boxes, int N,
                                                            const double *__restrict__ query, int K,
                                                            double *__restrict__ out) {
  __shared__ double q[64][5];
  const int k0 = blockIdx.y * 64, kn = min(64, K - k0);
  for (int i = threadIdx.x; i < kn; i += 256) {
    const double x1 = query[(long long)(k0 + i) * 4], y1 = query[(long long)(k0 + i) * 4 + 1];
    const double x2 = query[(long long)(k0 + i) * 4 + 2], y2 = query[(long long)(k0 + i) * 4 + 3];
    q[i][0] = x1; q[i][1] = y1; q[i][2] = x2; q[i][3] = y2;
    q[i][4] = (x2 - x1 + 1) * (y2 - y1 + 1);                      // box_area (bbox.pyx:35-38)
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const double bx1 = boxes[(long long)n * 4], by1 = boxes[(long long)n * 4 + 1];
  const double bx2 = boxes[(long long)n * 4 + 2], by2 = boxes[(long long)n * 4 + 3];
  const double barea = (bx2 - bx1 + 1) * (by2 - by1 + 1);
  double *o = out + (long long)n * K + k0;
  for (int k = 0; k < kn; ++k) {
    double v = 0.0;
    const double iw = fmin(bx2, q[k][2]) - fmax(bx1, q[k][0]) + 1;
    if (iw > 0) {
      const double ih = fmin(by2, q[k][3]) - fmax(by1, q[k][1]) + 1;
      if (ih > 0) {
        const double ua = barea + q[k][4] - iw * ih;
        v = iw * ih / ua;
      }
    }
    o[k] = v;
  }
}

}  // namespace

extern "C" {

size_t dspn_nms_pixel_workspace_bytes(int n) {
  if (n <= 0 || n > kMaxN) return 0;
  const size_t cb = (size_t)(n + 63) / 64;
  // order[n] ints | sorted boxes[n][4] floats | mask[n][cb] 64-bit words, each part 16-byte aligned
  return dspn::align_up(sizeof(int) * n, 16) + dspn::align_up(sizeof(float) * 4 * n, 16) + sizeof(unsigned long long) * n * cb;
}

int dspn_nms_pixel_f32(const float *dets_dev, int n, float thresh, int suppress_ge, int *keep_dev,
                       int *num_keep_dev, void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(keep_dev && num_keep_dev, "nms_pixel: null output pointer");
  DSPN_REQUIRE(suppress_ge >= 0 && suppress_ge <= 2, "nms_pixel: suppress_ge is 0 (nms), 1 (cpu_nms) or 2 (gpu_nms)");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) { (void)hipMemsetAsync(num_keep_dev, 0, sizeof(int), s); return 0; }
  DSPN_REQUIRE(dets_dev && n > 0 && n <= kMaxN, "nms_pixel: 1 <= n <= %d boxes", kMaxN);
  if (!workspace || workspace_bytes < dspn_nms_pixel_workspace_bytes(n))
    return dspn::fail(DSPN_ERR_WORKSPACE_, "nms_pixel: workspace too small");
  const int cb = (n + 63) / 64, np2 = pow2_at_least(n);
  char *w = static_cast<char *>(workspace);
  int *order = reinterpret_cast<int *>(w);
  float *sorted = reinterpret_cast<float *>(w + dspn::align_up(sizeof(int) * n, 16));
  unsigned long long *mask = reinterpret_cast<unsigned long long *>(
      w + dspn::align_up(sizeof(int) * n, 16) + dspn::align_up(sizeof(float) * 4 * n, 16));
  static dspn::KernelDeviceState st;
  if (const int dev = dspn::ensure_dynamic_lds(reinterpret_cast<const void *>(nms_sort_kernel), (size_t)kMaxN * 8, st, "nms_pixel"); dev < 0) return dev;
  hipLaunchKernelGGL(nms_sort_kernel, dim3(1), dim3(1024), (size_t)np2 * 8, s, dets_dev, n, np2, order, sorted);
  hipLaunchKernelGGL(nms_mask_kernel_px, dim3(cb, cb), dim3(64), 0, s, sorted, n, thresh, suppress_ge, mask, cb);
  hipLaunchKernelGGL(nms_scan_kernel_px, dim3(1), dim3(64), 0, s, mask, order, n, cb, keep_dev, num_keep_dev);
  return dspn::check_launch("nms_pixel");
}

int dspn_bbox_overlaps_f64(const double *boxes_dev, int N, const double *query_dev, int K, double *overlaps_dev,
                           void *stream) {
  DSPN_REQUIRE(N >= 0 && K >= 0, "bbox_overlaps: negative count");
  if (N == 0 || K == 0) return 0;
  DSPN_REQUIRE(boxes_dev && query_dev && overlaps_dev, "bbox_overlaps: null pointer");
  hipLaunchKernelGGL(bbox_overlaps_kernel, dim3((N + 255) / 256, (K + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     boxes_dev, N, query_dev, K, overlaps_dev);
  return dspn::check_launch("bbox_overlaps");
}

}  // extern "C"
