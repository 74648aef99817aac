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

// Per-DEVICE launch state of one kernel (advisor r5: a process-wide `static bool` cached the attribute of the first device
// for all of them and discarded the call's status): the dynamic-LDS limit is raised once per (kernel, device), the status is
// checked -- a kernel that asks for more LDS than the part has fails HERE, with a message, not at the launch -- and the
// persistent grids keep one occupancy figure per device.
constexpr int kMaxDevices = 32;
struct KernelDeviceState {
  bool ready[kMaxDevices] = {};
  int slots[kMaxDevices] = {}, slots_per_cu[kMaxDevices] = {}, cus[kMaxDevices] = {};
};
// -> device index (0 .. kMaxDevices - 1), or a negative dspn status with the message set
inline int ensure_dynamic_lds(const void *kern, size_t bytes, KernelDeviceState &st, const char *what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
    return fail(DSPN_ERR_LAUNCH_, "%s: no current device (or an index past %d)", what, kMaxDevices - 1);
  if (st.ready[dev]) return dev;
  if (bytes > 0) {
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess)
      return fail(DSPN_ERR_LAUNCH_, "%s: %zu bytes of dynamic LDS rejected on device %d (%s)", what, bytes, dev, hipGetErrorString(e));
  }
  st.ready[dev] = true;
  return dev;
}
// occupancy of a persistent kernel on the current device: workgroups per CU and CUs (cached per device in st)
inline int ensure_persistent_grid(const void *kern, int threads, size_t lds, KernelDeviceState &st, const char *what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
    return fail(DSPN_ERR_LAUNCH_, "%s: no current device (or an index past %d)", what, kMaxDevices - 1);
  if (st.slots[dev]) return dev;
  const int d2 = ensure_dynamic_lds(kern, lds, st, what);
  if (d2 < 0) return d2;
  int per_cu = 0, cus = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (per_cu < 1)
    return fail(DSPN_ERR_LAUNCH_, "%s: the kernel does not fit a compute unit of device %d (%d threads, %zu bytes of LDS)", what, dev, threads, lds);
  st.slots_per_cu[dev] = per_cu; st.cus[dev] = cus > 0 ? cus : 1;
  st.slots[dev] = per_cu * st.cus[dev] / 8 * 8 > 8 ? per_cu * st.cus[dev] / 8 * 8 : 8;
  return dev;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace dspn

#define DSPN_REQUIRE(cond, ...) \
  do { if (!(cond)) return dspn::fail(DSPN_ERR_ARG_, __VA_ARGS__); } while (0)
