// Shared host-side helpers of the dspnet_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#define DSPN_OK_ 0
#define DSPN_ERR_ARG_ (-1)
#define DSPN_ERR_WORKSPACE_ (-2)
#define DSPN_ERR_LAUNCH_ (-3)

namespace dspn {

char *last_error_buf();   // thread-local, 512 bytes (defined in capi.hip)

inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DSPN_ERR_LAUNCH_, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

bool prof_enabled();
void prof_begin(int family, hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s; bool on;
  ProfScope(int family, hipStream_t st) : s(st), on(prof_enabled()) { if (on) prof_begin(family, st); }
  ~ProfScope() { if (on) prof_end(s); }
};

// CUs the persistent convolution grids leave free (dspn_conv_set_reserved_cus; a launch setting, results do not depend on it)
int reserved_cus();
// tile family of the plane-fed two-piece convolutions (dspn_conv_set_wide_tiles; a launch setting as well)
int wide_tiles_mode();
// the tile-spanning loop of the short-K members of that family (dspn_conv_set_tile_spanning; round 6)
int tile_spanning();
// the batched form of the affine sampler's data gradient (dspn_affine_sampler_set_batched; round 6; same bits either way)
int sampler_batched();

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace dspn

#define DSPN_REQUIRE(cond, ...) \
  do { if (!(cond)) return dspn::fail(DSPN_ERR_ARG_, __VA_ARGS__); } while (0)
