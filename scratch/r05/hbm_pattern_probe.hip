// What HBM bandwidth does the A-operand access pattern of the 1x1 convolutions reach, without any arithmetic?
//   pattern 0: conv-like -- a workgroup of 256 threads owns 128 rows of a [M][C] float tensor (C = 256: 1 KiB per row) and walks
//              the channels in 8 steps of 32 (128 B per row and step: thread t reads 32 B of row t >> 2 (+64)), one step ahead
//   pattern 1: the same, two steps ahead
//   pattern 2: 256 B per row and step (4 steps), one ahead
//   pattern 3: row-contiguous: a wave reads whole rows (1 KiB per wave instruction), the workgroup 128 rows in 8 steps of 16 rows
//   pattern 4: plain streaming (grid-stride float4)
// build: hipcc -O3 --offload-arch=gfx950 hbm_pattern_probe.hip -o hbm_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PAT>
__global__ __launch_bounds__(256, 2) void probe(const float4 *__restrict__ x, float *__restrict__ out, int M, int C4, int ntiles) {
  const int tid = threadIdx.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add = [&](const float4 v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; };
  if (PAT == 4) {
    const long long n = (long long)M * C4;
    for (long long i = blockIdx.x * 256ll + tid; i < n; i += (long long)gridDim.x * 256) add(x[i]);
  } else {
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int m0 = t * 128;
      if (PAT == 0 || PAT == 1) {
        constexpr int D = PAT == 0 ? 1 : 2;
        float4 r[D + 1][4];
        const int row0 = m0 + (tid >> 2), q = (tid & 3) * 2;
        auto req = [&](int s, float4 (&dst)[4]) {
          const float4 *p0 = x + (size_t)row0 * C4 + s * 8 + q, *p1 = x + (size_t)(row0 + 64) * C4 + s * 8 + q;
          dst[0] = p0[0]; dst[1] = p0[1]; dst[2] = p1[0]; dst[3] = p1[1];
        };
        const int ns = C4 / 8;
#pragma unroll
        for (int d = 0; d < D; ++d) req(d, r[d]);
        for (int s = 0; s < ns; s += D + 1) {
#pragma unroll
          for (int u = 0; u < D + 1; ++u) {
            if (s + u >= ns) break;
            if (s + u + D < ns) req(s + u + D, r[(u + D) % (D + 1)]);
#pragma unroll
            for (int e = 0; e < 4; ++e) add(r[u][e]);
            __builtin_amdgcn_s_barrier();
          }
        }
      } else if (PAT == 2) {
        const int row0 = m0 + (tid >> 2), q = (tid & 3) * 4;
        float4 r[2][8];
        auto req = [&](int s, float4 (&dst)[8]) {
          const float4 *p0 = x + (size_t)row0 * C4 + s * 16 + q, *p1 = x + (size_t)(row0 + 64) * C4 + s * 16 + q;
#pragma unroll
          for (int e = 0; e < 4; ++e) { dst[e] = p0[e]; dst[4 + e] = p1[e]; }
        };
        const int ns = C4 / 16;
        req(0, r[0]);
        for (int s = 0; s < ns; s += 2) {
          if (s + 1 < ns) req(s + 1, r[1]);
#pragma unroll
          for (int e = 0; e < 8; ++e) add(r[0][e]);
          __builtin_amdgcn_s_barrier();
          if (s + 1 >= ns) break;
          if (s + 2 < ns) req(s + 2, r[0]);
#pragma unroll
          for (int e = 0; e < 8; ++e) add(r[1][e]);
          __builtin_amdgcn_s_barrier();
        }
      } else if (PAT == 3) {
        const int wave = tid >> 6, lane = tid & 63;
        float4 r[2][4];
        auto req = [&](int s, float4 (&dst)[4]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = x[(size_t)(m0 + s * 16 + wave * 4 + e) * C4 + lane];
        };
        req(0, r[0]);
        for (int s = 0; s < 8; s += 2) {
          req(s + 1, r[1]);
#pragma unroll
          for (int e = 0; e < 4; ++e) add(r[0][e]);
          __builtin_amdgcn_s_barrier();
          if (s + 2 < 8) req(s + 2, r[0]);
#pragma unroll
          for (int e = 0; e < 4; ++e) add(r[1][e]);
          __builtin_amdgcn_s_barrier();
        }
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e33f) out[0] = acc.x;
}

template <int PAT>
int run(const float4 *x, float *out, int M, int C) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int ntiles = M / 128;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(probe<PAT>, dim3(512), dim3(256), 0, 0, x, out, M, C / 4, ntiles);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep) printf("C %4d pattern %d: %7.1f us per pass, %6.2f TB/s\n", C, PAT, ms * 100.f, (double)M * C * 4 / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}

int main() {
  const int M = 524288;
  for (int C : {256, 512, 1024}) {
    if ((size_t)M * C * 4 > (size_t)3 << 30) break;
    float4 *x; float *out;
    CK(hipMalloc(&x, (size_t)M * C * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 0, (size_t)M * C * 4));
    if (C == 256) { if (run<0>(x, out, M, C) || run<1>(x, out, M, C) || run<2>(x, out, M, C) || run<3>(x, out, M, C)) return 1; }
    else { if (run<0>(x, out, M, C) || run<1>(x, out, M, C) || run<2>(x, out, M, C)) return 1; }
    if (run<4>(x, out, M, C)) return 1;
    CK(hipFree(x)); CK(hipFree(out));
  }
  return 0;
}
