// Round 6: the FINALIZE half of a BatchNorm backward -- per-channel sums of the row-tile table the data
// gradient's epilogue left, dgamma / dbeta, the three coefficients of dx = a dy' + c1 x + c0, the bound of dx -- as a JOB that
// rides in the next weight-gradient launch on the same stream instead of a launch of its own (bn_bwd_final_kernel: 1 - 32
// workgroups, 6 - 9 us during which the chip idles, 58 of them per training step).
//
// dspn_bn_backward_from_sums(flag | 2 | 8) launches tile_group_kernel<1> where the table is long (>= 1024 row tiles: that level
// needs hundreds of workgroups and stays a launch), fills a BnFinalJob for the rest and parks it (bn_job_defer); conv2d_wgrad_one takes it
// (bn_job_take) and adds rows of workgroups IN FRONT of its grid (blockIdx.y < job_rows: dispatched first) that run
// bn_final_job_run -- sixteen channels per workgroup, beside the weight gradient's own workgroups; the apply half
// (flag | 4) is launched behind that kernel by stream order.  If no weight gradient came by, the apply call launches the
// stand-alone form itself.  The sums are formed in bn_bwd_final_kernel's ORDER (sixteen double accumulators over the table rows,
// then 1 .. 15 onto 0): same bits.  (The first form of the job also summed the groups of a long table itself, sixteen channels per
// workgroup: 500 dependent loads per thread made it the LONGEST workgroup of a stage-1 weight gradient and its registers spilled
// the 128-register kernels -- the step lost 3.6 %.)
#pragma once
#include <hip/hip_runtime.h>

namespace dspn {

constexpr int kBnJobChannels = 16;     // channels per job workgroup

struct BnFinalJob {
  const float *tile_sums;              // [tiles][2][C]
  int tiles, C, blocks;                // blocks = ceil(C / 16) workgroups
  double inv_rows;
  const float *mean, *rstd, *gamma;
  float *coef, *dgamma, *dbeta;
  const float *dy_absmax, *x_minmax;
  unsigned *dx_bound, *dx_bound_min;
};

// host side (capi.hip): one parked job per stream
void bn_job_defer(hipStream_t s, const BnFinalJob &job);
bool bn_job_take(hipStream_t s, BnFinalJob *job);

// what bn_bwd_final_kernel does with a channel's two sums (shared with that kernel: one arithmetic)
__device__ __forceinline__ void bn_final_channel(const int c, const int C, const double S, const double SS, const double inv_rows,
                                                 const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                 float *__restrict__ coef, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                 const float *__restrict__ x_minmax, unsigned *__restrict__ dx_bound,
                                                 unsigned *__restrict__ dx_bound_min, const float D) {
  if (dbeta) dbeta[c] = (float)S;
  if (dgamma) dgamma[c] = (float)SS;
  const double rs = rstd[c], mu = mean[c];
  const double a = (gamma ? (double)gamma[c] : 1.0) * rs;
  const double c1 = -a * rs * (SS * inv_rows);
  coef[c] = (float)a;
  coef[C + c] = (float)c1;
  const double c0 = -a * (S * inv_rows) - c1 * mu;
  coef[2 * C + c] = (float)c0;
  if (dx_bound) {
    // |dx| = |a dy' + c1 x + c0| <= |a| D + max(|c1 lo + c0|, |c1 hi + c0|) over the channel's x in [lo, hi] (the affine
    // part is monotone in x): the magnitude block of the dx the apply kernel is ABOUT to write as piece planes -- a bound,
    // a few times the true maximum at most (the two-piece math tolerates 2^17)
    const double lo = x_minmax[c], hi = x_minmax[C + c];
    double b = fabs(a) * (double)D + fmax(fabs(c1 * lo + c0), fabs(c1 * hi + c0));
    b *= 1.0 + 1e-6;
    float bf = (float)b;
    if (!(bf == bf)) bf = INFINITY;
    if (bf > 0.f) atomicMax(dx_bound + (c & 63), __float_as_uint(bf));
    // round 5 (range guard): the SMALLEST non-zero per-channel bound -- the ratio to the block above is the span of channel
    // magnitudes the planes are cut over (one word, the caller presets it to +inf; min of positive floats = min of their bits)
    if (dx_bound_min && bf > 0.f && bf < INFINITY) atomicMin(dx_bound_min, __float_as_uint(bf));
  }
}

// one job workgroup: channels 16 b .. 16 b + 15 on threads 0 .. 255 (lane pair (channel, virtual slab lane 0 .. 15)); every
// thread of the workgroup reaches the barrier.  smem: 4 KiB + 4 bytes of the launch's dynamic LDS.
__device__ __forceinline__ void bn_final_job_run(const BnFinalJob &j, const int b, void *smem) {
  double *s_S = static_cast<double *>(smem), *s_SS = s_S + 16 * kBnJobChannels;
  float *s_D = reinterpret_cast<float *>(s_SS + 16 * kBnJobChannels);
  const int tid = threadIdx.x;
  if (j.dx_bound && tid < 64) {         // D = the largest |dy'| the producing data gradient stored (finite partial maxima)
    float m = j.dy_absmax[tid];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (tid == 0) *s_D = m;
  }
  const int cl = tid & (kBnJobChannels - 1), v = tid >> 4;
  const int c = b * kBnJobChannels + cl;
  const int C = j.C;
  if (tid < 256) {
    double S = 0, SS = 0;
    if (c < C) {
      const float *__restrict__ ts = j.tile_sums;
#pragma unroll 8
      for (int k = v; k < j.tiles; k += 16) {      // (bn_bwd_final_kernel's lane sl = v: the same terms in the same order)
        S += ts[(long long)k * 2 * C + c];
        SS += ts[(long long)k * 2 * C + C + c];
      }
    }
    s_S[v * kBnJobChannels + cl] = S; s_SS[v * kBnJobChannels + cl] = SS;
  }
  __syncthreads();
  if (tid < kBnJobChannels && c < C) {
    double S = s_S[cl], SS = s_SS[cl];
    for (int k = 1; k < 16; ++k) { S += s_S[k * kBnJobChannels + cl]; SS += s_SS[k * kBnJobChannels + cl]; }
    bn_final_channel(c, C, S, SS, j.inv_rows, j.mean, j.rstd, j.gamma, j.coef, j.dgamma, j.dbeta, j.x_minmax, j.dx_bound,
                     j.dx_bound_min, j.dx_bound ? *s_D : 0.f);
  }
}

}  // namespace dspn
