// Shared device/host helpers for libmi355pose (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mi355pose.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define WAVE 64

void mi355_set_error(const char* fmt, ...);
#define MI_FAIL(code, ...) do { mi355_set_error(__VA_ARGS__); return (code); } while (0)
#define MI_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
  if (e_ != hipSuccess) MI_FAIL(MI355_ELAUNCH, "%s: %s", (name), hipGetErrorString(e_)); } while (0)

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- element traits -------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
  __device__ static inline float ld(const float* p) { return *p; }
  __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPerChunk = 8;
  __device__ static inline float ld(const bf16_t* p) { return (float)*p; }
  __device__ static inline void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};

__device__ inline float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// 16-byte chunk <-> floats
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __device__ static inline void load(const float* p, float (&v)[4]) {
    float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
  __device__ static inline void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
  __device__ static inline void unpack(const uint4& q, float (&v)[4]) {
    v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w); }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  __device__ static inline void unpack(const uint4& q, float (&v)[8]) {
    unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  __device__ static inline void load(const bf16_t* p, float (&v)[8]) { unpack(*reinterpret_cast<const uint4*>(p), v); }
  __device__ static inline void store(bf16_t* p, const float (&v)[8]) {
    union { bf16_t h[8]; uint4 q; } u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u.h[i] = (bf16_t)v[i];
    *reinterpret_cast<uint4*>(p) = u.q;
  }
};

// CH consecutive bytes (CH = 8 or 4; the address is a multiple of CH) as ONE load / store: window-position bytes of the max-pool
template <int CH> __device__ inline void load_bytes(const uint8_t* p, unsigned (&b)[CH]) {
  if constexpr (CH == 8) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) { b[e] = (q.x >> (8 * e)) & 0xffu; b[4 + e] = (q.y >> (8 * e)) & 0xffu; }
  } else {
    const unsigned q = *reinterpret_cast<const unsigned*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = (q >> (8 * e)) & 0xffu;
  }
}
template <int CH> __device__ inline void store_bytes(uint8_t* p, const int (&b)[CH]) {
  if constexpr (CH == 8) {
    uint2 q; q.x = 0u; q.y = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) { q.x |= ((unsigned)b[e] & 0xffu) << (8 * e); q.y |= ((unsigned)b[4 + e] & 0xffu) << (8 * e); }
    *reinterpret_cast<uint2*>(p) = q;
  } else {
    unsigned q = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) q |= ((unsigned)b[e] & 0xffu) << (8 * e);
    *reinterpret_cast<unsigned*>(p) = q;
  }
}

// ---- wave / block reductions ----------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block of NW waves; result valid in all threads. `red` = shared float[NW].
template <int NW> __device__ inline float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += red[i];
  return r;
}
template <int NW> __device__ inline float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r = fmaxf(r, red[i]);
  return r;
}

// ---- exact unsigned division by a runtime constant (host-computed magic) ---------------
struct FastDiv {
  unsigned mul, shr, d;
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f; f.d = d;
  if (d == 1) { f.mul = 0; f.shr = 0; return f; }
  unsigned l = 0; while ((1ull << l) < d) ++l;          // ceil(log2 d)
  unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
  f.mul = (unsigned)m; f.shr = l; return f;
}
__device__ inline unsigned fd_div(unsigned n, const FastDiv& f) {
  if (f.d == 1) return n;
  unsigned t = __umulhi(n, f.mul);
  return (t + ((n - t) >> 1)) >> (f.shr - 1);
}

// ---- profiling (conv family) -----------------------------------------------------------
// family 0 = the MFMA implicit-GEMM conv kernels (what mi355_prof_read / _read_split sum: bench.py's roofline figure),
// family 1 = BatchNorm kernels, 2 = everything else that is logged.  Each scope brackets exactly ONE kernel launch; its label is
// the explicit one or, when that is null, the thread's pending tag (prof_set_tag: "fwd k3s1 256>256 @64x64 n64" ...), so that the
// launches of one iteration can be listed layer by layer (mi355_prof_read_launch) and joined, in launch order, with a rocprofv3
// kernel trace of the same iteration (profiles/insitu_table.py).
struct ProfScope {
  hipStream_t s; bool on; int slot;
  ProfScope(hipStream_t st, double flops, double bytes = 0.0, int family = 0, const char* label = nullptr);
  ~ProfScope();
};
bool prof_on();
void prof_set_tag(const char* fmt, ...);
