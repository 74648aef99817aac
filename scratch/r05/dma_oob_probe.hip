// Round 5 probe: what does `buffer_load_dwordx4 ... lds` leave in LDS (a) for a lane whose offset is out of the buffer's
// range, (b) for a lane that is masked off (EXEC = 0) while other lanes of the same instruction are active?
// Build: hipcc --offload-arch=gfx950 -O2 -o dma_oob_probe dma_oob_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float *src, unsigned bytes, float *out, int mode) {
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) smem[i] = -7.f;       // poison 2 KiB
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, bytes, 0x00020000);
  unsigned off = lane * 16u;
  if (mode == 0) { if (lane & 1) off |= 0x80000000u; }         // odd lanes out of range
  if (mode == 1) {
    if (lane & 1) {                                            // only odd lanes issue
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)smem, 16, off, 0, 0, 0);
    }
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)smem, 16, off, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = smem[i];
}
int main() {
  float *src, *out;
  hipMalloc(&src, 4096); hipMalloc(&out, 2048);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 1000.f + i;
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 2048, 0, src, 1024u, out, mode);
    std::vector<float> o(512);
    hipMemcpy(o.data(), out, 2048, hipMemcpyDeviceToHost);
    printf("mode %d (%s):\n", mode, mode == 0 ? "odd lanes out of range" : "only odd lanes active");
    for (int l = 0; l < 6; ++l) printf("  lane %d slot: %g %g %g %g\n", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    int zeros = 0, poison = 0, data = 0;
    for (int l = 0; l < 64; ++l) { const float v = o[4 * l]; if (v == 0.f) ++zeros; else if (v == -7.f) ++poison; else ++data; }
    printf("  slots: %d zero, %d untouched, %d data; beyond 1 KiB untouched: %s\n", zeros, poison, data, o[256] == -7.f ? "yes" : "NO");
  }
  return 0;
}
