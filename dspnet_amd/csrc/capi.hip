// Library-wide C ABI helpers (error string, version).
#include "dspn_common.h"
#include <vector>
#include "../../include/dspn_multibox.h"

namespace dspn {
char *last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace dspn

namespace dspn {
// Optional per-family kernel timing with HIP events recorded on the launch stream
// (bench.py's live roofline measurement).  Off by default: zero cost.
struct ProfRec { hipEvent_t a, b; int family; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
bool prof_enabled() { return g_prof_on; }
void prof_begin(int family, hipStream_t s) {
  ProfRec r; r.family = family;
  (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, s);
  g_prof.push_back(r);
}
void prof_end(hipStream_t s) { (void)hipEventRecord(g_prof.back().b, s); }
}  // namespace dspn

extern "C" {
int dspn_profile_enable(int on) { dspn::g_prof_on = on != 0; return 0; }
int dspn_profile_collect(int family, double *total_ms, long long *launches) {
  double t = 0; long long n = 0;
  std::vector<dspn::ProfRec> keep;
  for (auto &r : dspn::g_prof) {
    if (r.family != family) { keep.push_back(r); continue; }
    (void)hipEventSynchronize(r.b);
    float ms = 0; (void)hipEventElapsedTime(&ms, r.a, r.b);
    t += ms; ++n;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  dspn::g_prof.swap(keep);
  if (total_ms) *total_ms = t;
  if (launches) *launches = n;
  return 0;
}
const char *dspn_last_error(void) { return dspn::last_error_buf(); }
int dspn_abi_version(void) { return 1; }
}
