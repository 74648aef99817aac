// Round 6: what the chip gives a WRITE-dominated kernel (the N >> K 1x1 convolutions write 4 bytes per multiply-add column and
// read 1): fill / copy / read / the wide epilogue's store pattern (a tile of 128 rows x 512-byte row segments at a 1-KiB pitch).
//   hipcc --offload-arch=gfx950 -O3 -o scratch/r06/bw_probe scratch/r06/bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void fill_k(float4 *p, size_t n4) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void copy_k(const float4 *__restrict__ a, float4 *__restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = a[i];
}
__global__ __launch_bounds__(256) void read_k(const float4 *__restrict__ a, float *out, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 1.2345e33f) out[0] = s;
}
// tiles of 128 rows x 128 floats of a [rows][pitch] array, persistent workgroups; thread (er0 = tid / 32, c4 = tid % 32) stores rows er0 + 8 p
__global__ __launch_bounds__(256) void tile_store_k(float *p, int rows, int pitch, int ntiles, int ncol_tiles, int wait_each) {
  const int tid = threadIdx.x, c4 = tid & 31, er0 = tid >> 5;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int mt = t / ncol_tiles, nt = t - mt * ncol_tiles;
    float *base = p + (size_t)mt * 128 * pitch + nt * 128 + c4 * 4;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      *reinterpret_cast<float4 *>(base + (size_t)(er0 + 8 * q) * pitch) = make_float4((float)t, (float)q, 3.f, 4.f);
    if (wait_each) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  }
}
// read [rows][kp] (128 rows x 256 B per k-step, 2 k-steps) then store the tile: the short-K convolution without arithmetic
__global__ __launch_bounds__(256) void tile_rw_k(const float *__restrict__ a, float *p, int rows, int kp, int pitch, int ntiles, int ncol_tiles, int mode) {
  const int tid = threadIdx.x, c4 = tid & 31, er0 = tid >> 5;
  __shared__ float sm[256];
  float4 nxt[2];
  auto req = [&](int t, float4 *r) {
    const int mt = t / ncol_tiles;
    const float *src = a + (size_t)(mt * 128 + (tid >> 1)) * kp + (tid & 1) * 8;
    r[0] = *reinterpret_cast<const float4 *>(src); r[1] = *reinterpret_cast<const float4 *>(src + 4);
  };
  int t = blockIdx.x;
  if (mode == 1 && t < ntiles) req(t, nxt);
  for (; t < ntiles; t += gridDim.x) {
    float4 cur[2];
    if (mode == 1) { cur[0] = nxt[0]; cur[1] = nxt[1]; if (t + gridDim.x < ntiles) req(t + gridDim.x, nxt); }
    else req(t, cur);
    const float s = cur[0].x + cur[1].w;
    sm[tid] = s; __syncthreads();
    const float u = sm[(tid + 1) & 255]; __syncthreads();
    const int mt = t / ncol_tiles, nt = t - mt * ncol_tiles;
    float *base = p + (size_t)mt * 128 * pitch + nt * 128 + c4 * 4;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      *reinterpret_cast<float4 *>(base + (size_t)(er0 + 8 * q) * pitch) = make_float4(u, (float)q, 3.f, 4.f);
  }
}
template <typename F> static float timeit(F f, int reps = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const int rows = 524288, N = 256, K = 64;
  const size_t ne = (size_t)rows * N, n4 = ne / 4;
  float *p, *a, *o; CK(hipMalloc(&p, ne * 4)); CK(hipMalloc(&a, ne * 4)); CK(hipMalloc(&o, 1024));
  CK(hipMemset(a, 0, ne * 4));
  const double GB = ne * 4 / 1e9;
  for (int g : {1024, 2048, 4096, 16384}) {
    float t = timeit([&] { hipLaunchKernelGGL(fill_k, dim3(g), dim3(256), 0, 0, (float4 *)p, n4); });
    printf("fill  grid %5d: %7.1f us  %5.2f TB/s written\n", g, t * 1e3, GB / t);
    t = timeit([&] { hipLaunchKernelGGL(copy_k, dim3(g), dim3(256), 0, 0, (const float4 *)a, (float4 *)p, n4); });
    printf("copy  grid %5d: %7.1f us  %5.2f TB/s read + written (%5.2f each)\n", g, t * 1e3, 2 * GB / t, GB / t);
    t = timeit([&] { hipLaunchKernelGGL(read_k, dim3(g), dim3(256), 0, 0, (const float4 *)a, o, n4); });
    printf("read  grid %5d: %7.1f us  %5.2f TB/s read\n", g, t * 1e3, GB / t);
  }
  const int ntiles = rows / 128 * (N / 128);
  for (int g : {512, 1024, 2048}) for (int w : {0, 1}) {
    float t = timeit([&] { hipLaunchKernelGGL(tile_store_k, dim3(g), dim3(256), 0, 0, p, rows, N, ntiles, N / 128, w); });
    printf("tile stores (128 x 512 B at 1 KiB pitch) grid %4d, wait for the acknowledgement per tile %d: %7.1f us  %5.2f TB/s written\n", g, w, t * 1e3, GB / t);
  }
  for (int g : {512, 1024, 2048}) for (int m : {0, 1}) {
    float t = timeit([&] { hipLaunchKernelGGL(tile_rw_k, dim3(g), dim3(256), 0, 0, a, p, rows, K, N, ntiles, N / 128, m); });
    printf("tile read (128 x 256 B) + stores, grid %4d, next tile's rows requested before the stores %d: %7.1f us  %5.2f TB/s (algorithmic %.0f MB)\n",
           g, m, t * 1e3, (GB + rows * (double)K * 4 / 1e9) / t, (GB + rows * (double)K * 4 / 1e9) * 1e3);
  }
  return 0;
}
