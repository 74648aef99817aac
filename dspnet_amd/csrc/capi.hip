// Library-wide C ABI helpers (error string, version).
#include "dspn_common.h"
#include "bn_final_job.h"
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <utility>
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
// The one piece of process-wide state in the library: an opt-in measurement aid (bench.py), never touched by a compute
// entry point unless enabled.  Guarded by a mutex so that callers on several threads stay safe while it is on; each
// thread's open scope is tracked separately (prof_begin / prof_end pairs never interleave within a thread).
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;      // events handed back by dspn_profile_collect, reused by later scopes
static thread_local hipEvent_t t_prof_end = nullptr;
bool prof_enabled() { return g_prof_on.load(std::memory_order_relaxed); }
static hipEvent_t prof_event() {
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void prof_begin(int family, hipStream_t s) {
  ProfRec r; r.family = family;
  r.a = prof_event(); r.b = prof_event();      // creating an event costs microseconds of host time per launch: pooled
  (void)hipEventRecord(r.a, s);
  t_prof_end = r.b;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
}
void prof_end(hipStream_t s) { if (t_prof_end) (void)hipEventRecord(t_prof_end, s); t_prof_end = nullptr; }
// A launch setting, not compute state (results are bit-identical under every value): how many CUs' worth of workgroup slots
// the PERSISTENT convolution grids leave free for the kernels of other queues -- RCCL's all-reduce of the gradient buckets,
// which otherwise only finds room between two convolution launches.
static std::atomic<int> g_reserved_cus{0};
int reserved_cus() { return g_reserved_cus.load(std::memory_order_relaxed); }
static std::atomic<int> g_wide_tiles{0};
int wide_tiles_mode() { return g_wide_tiles.load(std::memory_order_relaxed); }
// round 6: the tile-spanning loop of the short-K members of the wide family (conv_wide.h, XT); DSPN_XT=0 starts with it off
static std::atomic<int> g_tile_spanning{[] { const char *e = getenv("DSPN_XT"); const int v = e ? atoi(e) : 1; return v < 0 ? 0 : (v > 2 ? 2 : v); }()};
int tile_spanning() { return g_tile_spanning.load(std::memory_order_relaxed); }
static std::atomic<int> g_sampler_batched{[] { const char *e = getenv("DSPN_SAMPLER_BATCHED"); return (e && atoi(e) == 0) ? 0 : 1; }()};
int sampler_batched() { return g_sampler_batched.load(std::memory_order_relaxed); }
// round 6: a BatchNorm-backward finalize parked for the next weight-gradient launch on its stream (bn_final_job.h).  One job
// per stream (a second one for the same stream replaces nothing: the first is handed to whoever asks first, then the second)
static std::mutex g_job_mu;
static std::vector<std::pair<hipStream_t, BnFinalJob>> g_jobs;
void bn_job_defer(hipStream_t s, const BnFinalJob &job) {
  std::lock_guard<std::mutex> lk(g_job_mu);
  g_jobs.emplace_back(s, job);
}
bool bn_job_take(hipStream_t s, BnFinalJob *job) {
  std::lock_guard<std::mutex> lk(g_job_mu);
  for (size_t i = 0; i < g_jobs.size(); ++i)
    if (g_jobs[i].first == s) {
      *job = g_jobs[i].second;
      g_jobs.erase(g_jobs.begin() + (long)i);
      return true;
    }
  return false;
}
}  // namespace dspn

extern "C" {
int dspn_profile_enable(int on) { dspn::g_prof_on.store(on != 0); return 0; }
int dspn_profile_collect(int family, double *total_ms, long long *launches) {
  double t = 0; long long n = 0;
  std::lock_guard<std::mutex> lk(dspn::g_prof_mu);
  std::vector<dspn::ProfRec> keep;
  for (auto &r : dspn::g_prof) {
    if (r.family != family) { keep.push_back(r); continue; }
    (void)hipEventSynchronize(r.b);
    float ms = 0; (void)hipEventElapsedTime(&ms, r.a, r.b);
    t += ms; ++n;
    dspn::g_prof_pool.push_back(r.a); dspn::g_prof_pool.push_back(r.b);
  }
  dspn::g_prof.swap(keep);
  if (total_ms) *total_ms = t;
  if (launches) *launches = n;
  return 0;
}
int dspn_conv_set_reserved_cus(int cus) {
  if (cus < 0 || cus > 128) return dspn::fail(DSPN_ERR_ARG_, "conv_set_reserved_cus: 0 .. 128 CUs, got %d", cus);
  dspn::g_reserved_cus.store(cus);
  return 0;
}
int dspn_conv_set_wide_tiles(int mode) {
  if (mode < 0 || mode > 4) return dspn::fail(DSPN_ERR_ARG_, "conv_set_wide_tiles: 0 automatic, 1 never, 2 256x128, 3 128x256, 4 128x128, got %d", mode);
  dspn::g_wide_tiles.store(mode);
  return 0;
}
int dspn_conv_set_tile_spanning(int on) {
  if (on < 0 || on > 2) return dspn::fail(DSPN_ERR_ARG_, "conv_set_tile_spanning: 0 off, 1 plane-fed kernels (default), 2 also the float-operand kernel, got %d", on);
  dspn::g_tile_spanning.store(on);
  return 0;
}
int dspn_affine_sampler_set_batched(int on) {
  if (on != 0 && on != 1) return dspn::fail(DSPN_ERR_ARG_, "affine_sampler_set_batched: 0 or 1, got %d", on);
  dspn::g_sampler_batched.store(on);
  return 0;
}
const char *dspn_last_error(void) { return dspn::last_error_buf(); }
int dspn_abi_version(void) { return 1; }
}
