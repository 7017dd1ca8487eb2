// Error reporting, version and the in-library kernel timer behind bench.py's roofline figure.
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";
void mi355_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* mi355_last_error(void) { return g_err; }
extern "C" int mi355_version(void) { return 100; }

// ---- profiling: hipEvent pairs around every MFMA-conv launch (only when enabled; never inside graph capture)
namespace {
struct ProfState {
  std::mutex mu; bool on = false;
  std::vector<hipEvent_t> ev;   // pairs
  size_t used = 0; double flops = 0, bytes = 0; long launches = 0;
};
ProfState& P() { static ProfState s; return s; }
}
ProfScope::ProfScope(hipStream_t st, double flops, double bytes) : s(st), on(false), slot(-1) {
  ProfState& p = P();
  if (!p.on) return;
  std::lock_guard<std::mutex> lk(p.mu);
  if (p.used + 2 > p.ev.size()) {
    for (int i = 0; i < 2; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; p.ev.push_back(e); }
  }
  slot = (int)p.used; p.used += 2; p.flops += flops; p.bytes += bytes; p.launches += 1; on = true;
  (void)hipEventRecord(p.ev[slot], s);
}
ProfScope::~ProfScope() {
  if (!on) return;
  ProfState& p = P();
  std::lock_guard<std::mutex> lk(p.mu);
  (void)hipEventRecord(p.ev[slot + 1], s);
}
extern "C" int mi355_prof_enable(int on) { P().on = on != 0; return MI355_OK; }
extern "C" int mi355_prof_reset(void) {
  ProfState& p = P(); std::lock_guard<std::mutex> lk(p.mu);
  p.used = 0; p.flops = 0; p.bytes = 0; p.launches = 0; return MI355_OK;
}
extern "C" int mi355_prof_read(double* total_ms, long* launches, double* flops, double* bytes) {
  ProfState& p = P(); std::lock_guard<std::mutex> lk(p.mu);
  double ms = 0;
  for (size_t i = 0; i + 1 < p.used; i += 2) {
    if (hipEventSynchronize(p.ev[i + 1]) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "prof: event sync failed");
    float t = 0; if (hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "prof: elapsed failed");
    ms += t;
  }
  if (total_ms) *total_ms = ms; if (launches) *launches = p.launches; if (flops) *flops = p.flops; if (bytes) *bytes = p.bytes;
  return MI355_OK;
}
