// Storage type of activation tensors (and of the convolution operands) in HBM.
//
// Every kernel file that touches activations is compiled twice: as is (float tensors, entry points `*_f32`) and
// through a two-line wrapper `<name>_h.hip` that defines DSPN_HALF and includes it again (bfloat16 tensors, entry
// points `*_bf16`).  Arithmetic is fp32 in both: a bf16 element is widened on load and rounded to nearest even on
// store.  Parameters, BatchNorm statistics, per-channel coefficient vectors, split-K slabs, loss inputs and the
// optimizer state are float in both builds.
//
// The proxies below give the bf16 build the `p[i]` / `p[i] = v` syntax of a float4 / float pointer, so one kernel body
// serves both storage types.
#pragma once
#include <hip/hip_runtime.h>
#define DSPN_BF16_ELEMENT __bf16   /* include/dspn_nn.h: dspn_bf16 is this type inside the library */

namespace dspn {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {   // round to nearest even (v_cvt_pk_bf16_f32)
  const bf16x2_t r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float4 widen4(u32x2_t w) {
  return make_float4(bf16_lo(w[0]), bf16_hi(w[0]), bf16_lo(w[1]), bf16_hi(w[1]));
}
__device__ __forceinline__ u32x2_t narrow4(float4 v) {
  u32x2_t r = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
  return r;
}
__device__ __forceinline__ float round_bf16(float v) { return (float)(__bf16)v; }

#ifdef DSPN_HALF
typedef __bf16 st_t;
constexpr bool kHalf = true;
#define DSPN_FN(base) base##_bf16
#define DSPN_FN_NAME(base) #base "_bf16"

struct A4Ref {
  u32x2_t *p;
  __device__ __forceinline__ operator float4() const { return widen4(*p); }
  __device__ __forceinline__ void operator=(float4 v) const { *p = narrow4(v); }
};
struct A4Ptr {    // 4 consecutive elements per index, 8-byte aligned
  u32x2_t *p;
  __host__ __device__ A4Ptr() : p(nullptr) {}
  __host__ __device__ explicit A4Ptr(void *q) : p(static_cast<u32x2_t *>(q)) {}
  __device__ __forceinline__ A4Ref operator[](long long i) const { return A4Ref{p + i}; }
  __device__ __forceinline__ A4Ptr operator+(long long i) const { A4Ptr r; r.p = p + i; return r; }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
struct CA4Ptr {
  const u32x2_t *p;
  __host__ __device__ CA4Ptr() : p(nullptr) {}
  __host__ __device__ explicit CA4Ptr(const void *q) : p(static_cast<const u32x2_t *>(q)) {}
  __host__ __device__ CA4Ptr(A4Ptr q) : p(q.p) {}
  __device__ __forceinline__ float4 operator[](long long i) const { return widen4(p[i]); }
  __device__ __forceinline__ CA4Ptr operator+(long long i) const { CA4Ptr r; r.p = p + i; return r; }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
struct A1Ref {
  __bf16 *p;
  __device__ __forceinline__ operator float() const { return (float)*p; }
  __device__ __forceinline__ void operator=(float v) const { *p = (__bf16)v; }
  __device__ __forceinline__ void operator+=(float v) const { *p = (__bf16)((float)*p + v); }
};
struct A1Ptr {
  __bf16 *p;
  __host__ __device__ A1Ptr() : p(nullptr) {}
  __host__ __device__ explicit A1Ptr(void *q) : p(static_cast<__bf16 *>(q)) {}
  __device__ __forceinline__ A1Ref operator[](long long i) const { return A1Ref{p + i}; }
  __device__ __forceinline__ A1Ptr operator+(long long i) const { A1Ptr r; r.p = p + i; return r; }
  __device__ __forceinline__ A4Ptr vec4() const { return A4Ptr(p); }   // caller guarantees 8-byte alignment
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
struct CA1Ptr {
  const __bf16 *p;
  __host__ __device__ CA1Ptr() : p(nullptr) {}
  __host__ __device__ explicit CA1Ptr(const void *q) : p(static_cast<const __bf16 *>(q)) {}
  __host__ __device__ CA1Ptr(A1Ptr q) : p(q.p) {}
  __device__ __forceinline__ float operator[](long long i) const { return (float)p[i]; }
  __device__ __forceinline__ CA1Ptr operator+(long long i) const { CA1Ptr r; r.p = p + i; return r; }
  __device__ __forceinline__ CA4Ptr vec4() const { return CA4Ptr(p); }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
#else
typedef float st_t;
constexpr bool kHalf = false;
#define DSPN_FN(base) base##_f32
#define DSPN_FN_NAME(base) #base "_f32"

struct A4Ptr {
  float4 *p;
  __host__ __device__ A4Ptr() : p(nullptr) {}
  __host__ __device__ explicit A4Ptr(void *q) : p(static_cast<float4 *>(q)) {}
  __device__ __forceinline__ float4 &operator[](long long i) const { return p[i]; }
  __device__ __forceinline__ A4Ptr operator+(long long i) const { A4Ptr r; r.p = p + i; return r; }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
struct CA4Ptr {
  const float4 *p;
  __host__ __device__ CA4Ptr() : p(nullptr) {}
  __host__ __device__ explicit CA4Ptr(const void *q) : p(static_cast<const float4 *>(q)) {}
  __host__ __device__ CA4Ptr(A4Ptr q) : p(q.p) {}
  __device__ __forceinline__ const float4 &operator[](long long i) const { return p[i]; }
  __device__ __forceinline__ CA4Ptr operator+(long long i) const { CA4Ptr r; r.p = p + i; return r; }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
struct A1Ptr {
  float *p;
  __host__ __device__ A1Ptr() : p(nullptr) {}
  __host__ __device__ explicit A1Ptr(void *q) : p(static_cast<float *>(q)) {}
  __device__ __forceinline__ float &operator[](long long i) const { return p[i]; }
  __device__ __forceinline__ A1Ptr operator+(long long i) const { A1Ptr r; r.p = p + i; return r; }
  __device__ __forceinline__ A4Ptr vec4() const { return A4Ptr(p); }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
struct CA1Ptr {
  const float *p;
  __host__ __device__ CA1Ptr() : p(nullptr) {}
  __host__ __device__ explicit CA1Ptr(const void *q) : p(static_cast<const float *>(q)) {}
  __host__ __device__ CA1Ptr(A1Ptr q) : p(q.p) {}
  __device__ __forceinline__ const float &operator[](long long i) const { return p[i]; }
  __device__ __forceinline__ CA1Ptr operator+(long long i) const { CA1Ptr r; r.p = p + i; return r; }
  __device__ __forceinline__ CA4Ptr vec4() const { return CA4Ptr(p); }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};
#endif

}  // namespace dspn
