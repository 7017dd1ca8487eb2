// Error reporting, version and the in-library kernel timer behind bench.py's roofline figure.
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <string>
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
  std::mutex mu; bool on = false; int level = 0;      // level 1: family 0 only, 2: every family
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<double> lflops, lbytes;   // per launch (slot / 2)
  std::vector<int> lfam; std::vector<std::string> llabel;
  size_t used = 0; double flops = 0, bytes = 0; long launches = 0;      // totals: family 0 only
};
ProfState& P() { static ProfState s; return s; }
thread_local char g_tag[160] = "";
}
bool prof_on() { return P().on; }
void prof_set_tag(const char* fmt, ...) {
  if (!P().on) return;
  va_list ap; va_start(ap, fmt); vsnprintf(g_tag, sizeof(g_tag), fmt, ap); va_end(ap);
}
ProfScope::ProfScope(hipStream_t st, double flops, double bytes, int family, const char* label) : s(st), on(false), slot(-1) {
  ProfState& p = P();
  if (!p.on || (family != 0 && p.level < 2)) return;
  std::lock_guard<std::mutex> lk(p.mu);
  if (p.used + 2 > p.ev.size()) {
    for (int i = 0; i < 2; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; p.ev.push_back(e); }
  }
  slot = (int)p.used; p.used += 2; on = true;
  if (family == 0) { p.flops += flops; p.bytes += bytes; p.launches += 1; }
  if (p.lflops.size() < p.used / 2) { p.lflops.resize(p.used / 2); p.lbytes.resize(p.used / 2); p.lfam.resize(p.used / 2); p.llabel.resize(p.used / 2); }
  p.lflops[slot / 2] = flops; p.lbytes[slot / 2] = bytes; p.lfam[slot / 2] = family;
  p.llabel[slot / 2] = label ? label : g_tag;
  if (!label) g_tag[0] = 0;
  (void)hipEventRecord(p.ev[slot], s);
}
ProfScope::~ProfScope() {
  if (!on) return;
  ProfState& p = P();
  std::lock_guard<std::mutex> lk(p.mu);
  (void)hipEventRecord(p.ev[slot + 1], s);
}
// on: 0 off, 1 the conv family only (bench.py's roofline pass), 2 every logged family (the in-situ layer table)
extern "C" int mi355_prof_enable(int on) { P().on = on != 0; P().level = on; return MI355_OK; }
extern "C" int mi355_prof_reset(void) {
  ProfState& p = P(); std::lock_guard<std::mutex> lk(p.mu);
  p.used = 0; p.flops = 0; p.bytes = 0; p.launches = 0; return MI355_OK;
}
extern "C" int mi355_prof_read(double* total_ms, long* launches, double* flops, double* bytes) {
  ProfState& p = P(); std::lock_guard<std::mutex> lk(p.mu);
  double ms = 0;
  for (size_t i = 0; i + 1 < p.used; i += 2) {
    if (p.lfam[i / 2] != 0) continue;
    if (hipEventSynchronize(p.ev[i + 1]) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "prof: event sync failed");
    float t = 0; if (hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "prof: elapsed failed");
    ms += t;
  }
  if (total_ms) *total_ms = ms; if (launches) *launches = p.launches; if (flops) *flops = p.flops; if (bytes) *bytes = p.bytes;
  return MI355_OK;
}

// One logged launch: its family (ProfScope), event-timed duration, algorithmic FLOPs / bytes and label, in launch order.
extern "C" int mi355_prof_launch_count(long* n) {
  if (!n) MI_FAIL(MI355_EINVAL, "prof_launch_count: n is null");
  ProfState& p = P(); std::lock_guard<std::mutex> lk(p.mu);
  *n = (long)(p.used / 2);
  return MI355_OK;
}
extern "C" int mi355_prof_read_launch(long i, int* family, double* us, double* flops, double* bytes, char* label, int label_cap) {
  ProfState& p = P(); std::lock_guard<std::mutex> lk(p.mu);
  if (i < 0 || (size_t)i >= p.used / 2) MI_FAIL(MI355_EINVAL, "prof_read_launch: index %ld of %zu", i, p.used / 2);
  if (hipEventSynchronize(p.ev[2 * i + 1]) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "prof: event sync failed");
  float t = 0; if (hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "prof: elapsed failed");
  if (family) *family = p.lfam[i];
  if (us) *us = t * 1e3; if (flops) *flops = p.lflops[i]; if (bytes) *bytes = p.lbytes[i];
  if (label && label_cap > 0) { snprintf(label, (size_t)label_cap, "%s", p.llabel[i].c_str()); }
  return MI355_OK;
}

// ---- a timed idle spin on the stream: lets a measurement enqueue its launches behind a known delay, so that the
// events bracketing each launch are processed back to back on the GPU instead of waiting for the host (bench.py)
__global__ void spin_kernel(long ticks) {
  const long t0 = (long)wall_clock64();                 // constant 100 MHz counter
  while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int mi355_spin_us(long us, void* stream) {
  if (us < 0 || us > 2000000) MI_FAIL(MI355_EINVAL, "spin_us: 0..2e6 us");
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, as_stream(stream), us * 100);
  MI_CHECK_LAUNCH("spin_us");
  return MI355_OK;
}

// What an event pair reads around a kernel that does nothing: the dispatch latency every event-timed launch carries on
// top of its own duration (rocprofv3 kernel durations do not contain it).  The n launches are queued behind a spin.
extern "C" int mi355_prof_event_overhead_us(int n, void* stream, double* us) {
  if (n < 1 || n > 4096 || !us) MI_FAIL(MI355_EINVAL, "prof_event_overhead_us: bad args");
  hipStream_t st = as_stream(stream);
  std::vector<hipEvent_t> ev(2 * n);
  for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "event create failed");
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, st, 20000L * 100);      // 20 ms head start for the host
  for (int i = 0; i < n; ++i) {
    (void)hipEventRecord(ev[2 * i], st);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, st, 0L);
    (void)hipEventRecord(ev[2 * i + 1], st);
  }
  if (hipStreamSynchronize(st) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "sync failed");
  double tot = 0;
  for (int i = 0; i < n; ++i) { float t = 0; (void)hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]); tot += t; }
  for (auto& e : ev) (void)hipEventDestroy(e);
  *us = tot * 1e3 / n;
  return MI355_OK;
}

// The event-timed launches split by arithmetic intensity (algorithmic FLOP per algorithmic byte): launches below
// `flop_per_byte` are priced against HBM, the others against MFMA.  out[0..3] = {ms, flops, bytes, launches} of the
// low-intensity class, out[4..7] of the high-intensity class.
extern "C" int mi355_prof_read_split(double flop_per_byte, double* out) {
  if (!out) MI_FAIL(MI355_EINVAL, "prof_read_split: out missing");
  ProfState& p = P(); std::lock_guard<std::mutex> lk(p.mu);
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  for (size_t i = 0; i + 1 < p.used; i += 2) {
    if (p.lfam[i / 2] != 0) continue;
    if (hipEventSynchronize(p.ev[i + 1]) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "prof: event sync failed");
    float t = 0; if (hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "prof: elapsed failed");
    const double f = p.lflops[i / 2], b = p.lbytes[i / 2];
    const int c = (b > 0 && f / b < flop_per_byte) ? 0 : 4;
    out[c] += t; out[c + 1] += f; out[c + 2] += b; out[c + 3] += 1.0;
  }
  return MI355_OK;
}
