// Heat-map row kernels (rows = B*K maps of H*W fp32): hard arg-max (bit-exact with numpy), soft-arg-max,
// fused log-softmax + KL loss (+ its gradient), pseudo-label builders, bilinear up-sampling, PCK distances.
// One 256-thread workgroup per map, wave-shuffle reductions; all HBM/L2-bound.
#include "common.h"
#include <stdlib.h>

struct ArgBest { float v; int i; };
__device__ __forceinline__ bool arg_better(float av, int ai, float bv, int bi) {
  const bool an = av != av, bn = bv != bv;
  if (an || bn) return (an && !bn) || (an && bn && ai < bi);
  return av > bv || (av == bv && ai < bi);
}

// utils/keypoint_detection.py:7-35
__global__ __launch_bounds__(256) void argmax2d_kernel(const float* __restrict__ hm, int* __restrict__ idx, float* __restrict__ xy,
                                                        float* __restrict__ maxval, int HW, int W) {
  __shared__ float sv[4]; __shared__ int si[4];
  const float* row = hm + (size_t)blockIdx.x * HW;
  float bv = row[0]; int bi = 0;
  if (threadIdx.x < HW) { bv = row[threadIdx.x]; bi = threadIdx.x; }
  for (int i = threadIdx.x + 256; i < HW; i += 256) { const float v = row[i]; if (arg_better(v, i, bv, bi)) { bv = v; bi = i; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) if (arg_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    const bool pos = bv > 0.0f;                       // np.greater(maxvals, 0.0): NaN -> False
    if (idx) idx[blockIdx.x] = bi;
    if (maxval) maxval[blockIdx.x] = bv;
    if (xy) { xy[2 * blockIdx.x] = pos ? (float)(bi % W) : 0.f; xy[2 * blockIdx.x + 1] = pos ? (float)(bi / W) : 0.f; }
  }
}

// utils/keypoint_detection.py:209-239
__global__ __launch_bounds__(256) void softargmax_kernel(const float* __restrict__ hm, float* __restrict__ uv, int HW, int W, float beta,
                                                          float out_scale) {
  __shared__ float red[4];
  const float* row = hm + (size_t)blockIdx.x * HW;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < HW; i += 256) m = fmaxf(m, row[i] * beta);
  m = block_max<4>(m, red);
  float s = 0.f, su = 0.f, sv = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float e = expf(row[i] * beta - m);
    s += e; su += e * (float)(i % W); sv += e * (float)(i / W);
  }
  s = block_sum<4>(s, red); su = block_sum<4>(su, red); sv = block_sum<4>(sv, red);
  if (threadIdx.x == 0) { uv[2 * blockIdx.x] = su / s * out_scale; uv[2 * blockIdx.x + 1] = sv / s * out_scale; }
}

// uda/model/loss.py:145-158 (+ d(mean loss)/d pred)
__global__ __launch_bounds__(256) void kl_heatmap_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                          const float* __restrict__ weight, float eps, float* __restrict__ loss_rows,
                                                          float* __restrict__ unit_grad, int HW, float inv_count) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * HW;
  const float* p = pred + base; const float* t = target + base;
  float m = -INFINITY, ts = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) { m = fmaxf(m, p[i]); ts += t[i] + eps; }
  m = block_max<4>(m, red);
  ts = block_sum<4>(ts, red);
  float se = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) se += expf(p[i] - m);
  se = block_sum<4>(se, red);
  const float lse = logf(se);
  float l = 0.f, tn_sum = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float tn = (t[i] + eps) / ts;
    const float logp = p[i] - m - lse;
    // nn.KLDivLoss pointwise: xlogy(t,t) - t*logp  (xlogy(0,0) = 0; NaN propagates)
    const float xl = (tn == 0.f) ? 0.f : tn * logf(tn);
    l += xl - tn * logp;
    tn_sum += tn;
  }
  l = block_sum<4>(l, red);
  tn_sum = block_sum<4>(tn_sum, red);
  const float w = weight ? weight[blockIdx.x] : 1.f;
  if (threadIdx.x == 0) loss_rows[blockIdx.x] = l * w;
  if (unit_grad) {
    float* g = unit_grad + base;
    const float k = w * inv_count;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float tn = (t[i] + eps) / ts;
      const float sm = expf(p[i] - m - lse);
      g[i] = k * (sm * tn_sum - tn);
    }
  }
}

// ---------------------------------------------------------------------------------------- register-resident row kernels
// The three row kernels above walk a map two to four times with 4-byte loads and a block reduction between the walks: at one
// 16-KB map per workgroup they are latency-bound (arg-max 1.8 TB/s, soft-arg-max and KL 1.3 - 2 TB/s of the 22-MB heat-map
// tensors by counters, round 2).  These forms fetch a map ONCE with 16-byte loads, all of them in flight together, and keep it
// in registers for every pass:
//   WPM = false: one 256-thread workgroup per map of 1024 * NV floats (64 x 64 maps: NV = 4), block reductions through LDS;
//   WPM = true : one WAVE per map of 256 * NV floats (16 x 16: NV = 1, 32 x 32: NV = 4), four maps per workgroup, reductions by
//                wave shuffles only -- no LDS, no barrier.
// Same arithmetic per element as the kernels above; sums are folded in a different (fixed) order.
template <int NV, bool WPM> struct RowGeom {
  static constexpr int kThreads = WPM ? 64 : 256;                         // threads per map
  __device__ static int map() { return WPM ? (int)(blockIdx.x * 4 + (threadIdx.x >> 6)) : (int)blockIdx.x; }
  __device__ static int tid() { return WPM ? (int)(threadIdx.x & 63) : (int)threadIdx.x; }
  __device__ static float sum(float v, float* red) { if constexpr (WPM) return wave_sum(v); else return block_sum<4>(v, red); }
  __device__ static float max(float v, float* red) { if constexpr (WPM) return wave_max(v); else return block_max<4>(v, red); }
  // three sums with ONE exchange through LDS (red3: 12 floats) instead of three barrier pairs
  __device__ static void sum3(float& a, float& b, float& c, float* red3) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if constexpr (!WPM) {
      __syncthreads();
      if ((threadIdx.x & 63) == 0) { float* q = red3 + 3 * (threadIdx.x >> 6); q[0] = a; q[1] = b; q[2] = c; }
      __syncthreads();
      a = (red3[0] + red3[3]) + (red3[6] + red3[9]); b = (red3[1] + red3[4]) + (red3[7] + red3[10]); c = (red3[2] + red3[5]) + (red3[8] + red3[11]);
    }
  }
};
template <int NV, bool WPM>
__device__ __forceinline__ void row_load(const float* __restrict__ row, float (&v)[4 * NV]) {
  using G = RowGeom<NV, WPM>;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const float4 q = reinterpret_cast<const float4*>(row)[G::tid() + G::kThreads * j];
    v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
  }
}
// element index of register slot (j, e) of this thread
template <int NV, bool WPM> __device__ __forceinline__ int row_index(int j, int e) { return 4 * (RowGeom<NV, WPM>::tid() + RowGeom<NV, WPM>::kThreads * j) + e; }

template <int NV, bool WPM>
__global__ __launch_bounds__(256) void argmax2d_reg_kernel(const float* __restrict__ hm, int* __restrict__ idx, float* __restrict__ xy,
                                                            float* __restrict__ maxval, int rows, int W) {
  using G = RowGeom<NV, WPM>;
  constexpr int HW = 4 * NV * G::kThreads;
  __shared__ float sv[4]; __shared__ int si[4];
  const int m = G::map();
  if (WPM && m >= rows) return;                              // (wave-uniform: a whole wave owns a map)
  float v[4 * NV];
  row_load<NV, WPM>(hm + (size_t)m * HW, v);
  float bv = v[0]; int bi = row_index<NV, WPM>(0, 0);
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int i = row_index<NV, WPM>(j, e); if (arg_better(v[4 * j + e], i, bv, bi)) { bv = v[4 * j + e]; bi = i; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (arg_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if constexpr (!WPM) {
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int w = 1; w < 4; ++w) if (arg_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
  }
  if (G::tid() == 0) {
    const bool pos = bv > 0.0f;                              // np.greater(maxvals, 0.0): NaN -> False
    if (idx) idx[m] = bi;
    if (maxval) maxval[m] = bv;
    if (xy) { xy[2 * m] = pos ? (float)(bi % W) : 0.f; xy[2 * m + 1] = pos ? (float)(bi / W) : 0.f; }
  }
}

template <int NV, bool WPM>
__global__ __launch_bounds__(256) void softargmax_reg_kernel(const float* __restrict__ hm, float* __restrict__ uv, int rows, int W, float beta,
                                                              float out_scale) {
  using G = RowGeom<NV, WPM>;
  constexpr int HW = 4 * NV * G::kThreads;
  __shared__ float red[4];
  const int m = G::map();
  if (WPM && m >= rows) return;
  float v[4 * NV];
  row_load<NV, WPM>(hm + (size_t)m * HW, v);
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4 * NV; ++k) mx = fmaxf(mx, v[k] * beta);
  mx = G::max(mx, red);
  float s = 0.f, su = 0.f, sv = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = row_index<NV, WPM>(j, e);
      const float ex = expf(v[4 * j + e] * beta - mx);
      s += ex; su += ex * (float)(i % W); sv += ex * (float)(i / W);
    }
  __shared__ float red3[12];
  G::sum3(s, su, sv, red3);
  if (G::tid() == 0) { uv[2 * m] = su / s * out_scale; uv[2 * m + 1] = sv / s * out_scale; }
}

template <int NV, bool WPM>
__global__ __launch_bounds__(256) void kl_heatmap_reg_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                              const float* __restrict__ weight, float eps, float* __restrict__ loss_rows,
                                                              float* __restrict__ unit_grad, int rows, float inv_count) {
  using G = RowGeom<NV, WPM>;
  constexpr int HW = 4 * NV * G::kThreads;
  __shared__ float red[4];
  const int m = G::map();
  if (WPM && m >= rows) return;
  const size_t base = (size_t)m * HW;
  float p[4 * NV], t[4 * NV];
  row_load<NV, WPM>(pred + base, p);
  row_load<NV, WPM>(target + base, t);
  float mx = -INFINITY, ts = 0.f;
#pragma unroll
  for (int k = 0; k < 4 * NV; ++k) { mx = fmaxf(mx, p[k]); ts += t[k] + eps; }
  mx = G::max(mx, red);
  ts = G::sum(ts, red);
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < 4 * NV; ++k) se += expf(p[k] - mx);
  se = G::sum(se, red);
  const float lse = logf(se);
  float l = 0.f, tn_sum = 0.f;
#pragma unroll
  for (int k = 0; k < 4 * NV; ++k) {
    const float tn = (t[k] + eps) / ts;
    const float logp = p[k] - mx - lse;
    const float xl = (tn == 0.f) ? 0.f : tn * logf(tn);      // nn.KLDivLoss pointwise: xlogy(t,t) - t*logp
    l += xl - tn * logp;
    tn_sum += tn;
    t[k] = tn;                                               // (kept for the gradient)
  }
  l = G::sum(l, red);
  tn_sum = G::sum(tn_sum, red);
  const float w = weight ? weight[m] : 1.f;
  if (G::tid() == 0) loss_rows[m] = l * w;
  if (unit_grad) {
    const float kk = w * inv_count;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float4 g;
      g.x = kk * (expf(p[4 * j] - mx - lse) * tn_sum - t[4 * j]);
      g.y = kk * (expf(p[4 * j + 1] - mx - lse) * tn_sum - t[4 * j + 1]);
      g.z = kk * (expf(p[4 * j + 2] - mx - lse) * tn_sum - t[4 * j + 2]);
      g.w = kk * (expf(p[4 * j + 3] - mx - lse) * tn_sum - t[4 * j + 3]);
      reinterpret_cast<float4*>(unit_grad + base)[G::tid() + G::kThreads * j] = g;
    }
  }
}

// row-kernel dispatch: 0 = no register form for this map size (or unaligned pointers): the loop kernels above
static int row_form(int HW, const void* a, const void* b = nullptr, const void* c = nullptr) {
  static const bool on = !(getenv("MI355_ROW_REG") && atoi(getenv("MI355_ROW_REG")) == 0);      // A/B switch
  if (!on) return 0;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) return 0;
  if (HW == 4096) return 1;       // block per map, 16 floats per thread (a wave per 64 x 64 map, 64 floats per lane, measured slower:
                                  // arg-max 9.5 vs 6.5 us, soft-arg-max 12.4 vs 8.7, KL 22.5 vs 18.1 on the 64 x 21 x 64 x 64 tensor)
  if (HW == 1024) return 2;       // wave per map, 16 floats per lane
  if (HW == 256) return 3;        // wave per map, 4 floats per lane
  return 0;
}

__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int n, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += in[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

__global__ void scale_by_dev_kernel(const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ out, long n) {
  const float k = *g;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i] * k;
}

// out = in * (*g) on a feature tensor of n elements (n % 8 == 0 for bf16, % 4 for fp32): 16-byte accesses.
template <typename T>
__global__ __launch_bounds__(256) void scale_feature_kernel(const T* __restrict__ in, const float* __restrict__ g, T* __restrict__ out, long nvec) {
  constexpr int PER = Chunk<T>::N;
  const float k = *g;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float v[PER];
    Chunk<T>::load(in + i * PER, v);
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] *= k;
    Chunk<T>::store(out + i * PER, v);
  }
}

// Pseudo labels (see mi355pose.h).  One block per (b,k) map.
__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ xy, const float* __restrict__ patch, int radius, int div,
                                                            int S, int kind, const float* __restrict__ extra, int normalise,
                                                            float* __restrict__ gt, float* __restrict__ gf, int K) {
  __shared__ int cx[64], cy[64];
  __shared__ float spatch[13 * 13];
  __shared__ float red[4];
  const int b = blockIdx.x / K, k = blockIdx.x % K;
  const int side = 2 * radius + 1;
  if (threadIdx.x < K) {
    const float x = xy[2 * (b * K + threadIdx.x)], y = xy[2 * (b * K + threadIdx.x) + 1];
    cx[threadIdx.x] = (int)(x / (float)div); cy[threadIdx.x] = (int)(y / (float)div);   // .astype(int): truncation
  }
  for (int i = threadIdx.x; i < side * side; i += 256) spatch[i] = patch[i];
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * S * S;
  auto gauss = [&](int j, int px, int py) -> float {
    const int dx = px - cx[j] + radius, dy = py - cy[j] + radius;
    return (dx >= 0 && dx < side && dy >= 0 && dy < side) ? spatch[dy * side + dx] : 0.f;
  };
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < S * S; i += 256) {
    const int px = i % S, py = i / S;
    const float g = gauss(k, px, py);
    if (gt) gt[base + i] = g;
    if (gf) {
      float f;
      if (kind == 1) f = fminf(fmaxf(1.f - g * 10.f, 0.f), 1.f);
      else {
        float s = 0.f;
        if (kind == 0) { for (int j = 0; j < K; ++j) if (j != k) s += gauss(j, px, py); f = fminf(fmaxf(s, 0.f), 1.f); }
        else { for (int j = 0; j < K; ++j) s += gauss(j, px, py); f = fminf(fmaxf(fminf(fmaxf(s, 0.f), 1.f) - g * 10.f, 0.f), 1.f); }
      }
      if (extra) f = fminf(fmaxf(f + extra[base + i] - g * 100.f, 0.f), 1.f);
      gf[base + i] = f;
      mx = fmaxf(mx, f);     // fmaxf drops NaN like torch.max? torch.max propagates NaN; extra is finite in practice
    }
  }
  if (gf && normalise) {
    mx = block_max<4>(mx, red);
    __syncthreads();
    if (normalise == 2 && !(mx > 0.f)) return;          // guarded mode: an all-zero map stays zero (block-uniform)
    for (int i = threadIdx.x; i < S * S; i += 256) gf[base + i] = gf[base + i] / mx;   // 0/0 -> NaN as the reference
  }
}

// register-resident form of pseudo_label_kernel for maps of 1024 * NV pixels (64 x 64: NV = 4, 32 x 32: NV = 1): the ground-false
// values stay in registers between the clip and the per-map max-normalisation (the loop form writes them, then reads them back
// and writes them again: 1.38x its algorithmic traffic by counters), every store is 16 bytes.  Same arithmetic per pixel.
template <int NV>
__global__ __launch_bounds__(256) void pseudo_label_reg_kernel(const float* __restrict__ xy, const float* __restrict__ patch, int radius, int div,
                                                                int S, int kind, const float* __restrict__ extra, int normalise,
                                                                float* __restrict__ gt, float* __restrict__ gf, int K) {
  __shared__ int cx[64], cy[64];
  __shared__ float spatch[13 * 13];
  __shared__ float red[4];
  const int b = blockIdx.x / K, k = blockIdx.x % K;
  const int side = 2 * radius + 1;
  if (threadIdx.x < K) {
    const float x = xy[2 * (b * K + threadIdx.x)], y = xy[2 * (b * K + threadIdx.x) + 1];
    cx[threadIdx.x] = (int)(x / (float)div); cy[threadIdx.x] = (int)(y / (float)div);   // .astype(int): truncation
  }
  for (int i = threadIdx.x; i < side * side; i += 256) spatch[i] = patch[i];
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * S * S;
  auto gauss = [&](int j, int px, int py) -> float {
    const int dx = px - cx[j] + radius, dy = py - cy[j] + radius;
    return (dx >= 0 && dx < side && dy >= 0 && dy < side) ? spatch[dy * side + dx] : 0.f;
  };
  float f[4 * NV];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v4 = threadIdx.x + 256 * j;
    float4 ex = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gf && extra) ex = reinterpret_cast<const float4*>(extra + base)[v4];
    const float exv[4] = {ex.x, ex.y, ex.z, ex.w};
    float g4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * v4 + e, px = i % S, py = i / S;
      const float g = gauss(k, px, py);
      g4[e] = g;
      float fv = 0.f;
      if (gf) {
        if (kind == 1) fv = fminf(fmaxf(1.f - g * 10.f, 0.f), 1.f);
        else {
          float s = 0.f;
          if (kind == 0) { for (int q = 0; q < K; ++q) if (q != k) s += gauss(q, px, py); fv = fminf(fmaxf(s, 0.f), 1.f); }
          else { for (int q = 0; q < K; ++q) s += gauss(q, px, py); fv = fminf(fmaxf(fminf(fmaxf(s, 0.f), 1.f) - g * 10.f, 0.f), 1.f); }
        }
        if (extra) fv = fminf(fmaxf(fv + exv[e] - g * 100.f, 0.f), 1.f);
        mx = fmaxf(mx, fv);
      }
      f[4 * j + e] = fv;
    }
    if (gt) reinterpret_cast<float4*>(gt + base)[v4] = make_float4(g4[0], g4[1], g4[2], g4[3]);
  }
  if (!gf) return;
  bool divide = false;
  if (normalise) {
    mx = block_max<4>(mx, red);
    divide = !(normalise == 2 && !(mx > 0.f));               // guarded mode: an all-zero map stays zero (block-uniform)
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float4 o;
    o.x = divide ? f[4 * j] / mx : f[4 * j]; o.y = divide ? f[4 * j + 1] / mx : f[4 * j + 1];      // 0/0 -> NaN as the reference
    o.z = divide ? f[4 * j + 2] / mx : f[4 * j + 2]; o.w = divide ? f[4 * j + 3] / mx : f[4 * j + 3];
    reinterpret_cast<float4*>(gf + base)[threadIdx.x + 256 * j] = o;
  }
}

// nn.Upsample(size, mode='bilinear'), align_corners=False (ATen upsample_bilinear2d index rule)
__global__ void bilinear_up_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int h, int w, int H, int W,
                                   float alpha, int accumulate) {
  const long total = (long)rows * H * W;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(id % W); long r = id / W; const int oy = (int)(r % H); const int row = (int)(r / H);
    float fy = sy * ((float)oy + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = sx * ((float)ox + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < h - 1) ? 1 : 0), x1 = x0 + ((x0 < w - 1) ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* src = in + (size_t)row * h * w;
    const float v = hy * (hx * src[y0 * w + x0] + lx * src[y0 * w + x1]) + ly * (hx * src[y1 * w + x0] + lx * src[y1 * w + x1]);
    out[id] = alpha * v + (accumulate ? out[id] : 0.f);
  }
}

// utils/keypoint_detection.py:38-50 (float64 arithmetic like numpy)
__global__ void pck_dists_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, float* __restrict__ dists, int rows,
                                 float nx, float ny) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float tx = tgt[2 * r], ty = tgt[2 * r + 1];
  float d = -1.f;
  if (tx > 1.f && ty > 1.f) {
    const double dx = (double)pred[2 * r] / (double)nx - (double)tx / (double)nx;
    const double dy = (double)pred[2 * r + 1] / (double)ny - (double)ty / (double)ny;
    d = (float)sqrt(dx * dx + dy * dy);
  }
  dists[r] = d;
}

// ---------------------------------------------------------------------------------------- host
extern "C" int mi355_argmax2d(const float* hm, int32_t* idx, float* xy, float* maxval, int rows, int H, int W, void* stream) {
  if (!hm || rows < 1 || H < 1 || W < 1) MI_FAIL(MI355_EINVAL, "argmax2d: bad args");
  hipStream_t st = as_stream(stream);
  switch (row_form(H * W, hm)) {
    case 1: hipLaunchKernelGGL((argmax2d_reg_kernel<4, false>), dim3(rows), dim3(256), 0, st, hm, idx, xy, maxval, rows, W); break;
    case 2: hipLaunchKernelGGL((argmax2d_reg_kernel<4, true>), dim3(cdiv(rows, 4)), dim3(256), 0, st, hm, idx, xy, maxval, rows, W); break;
    case 3: hipLaunchKernelGGL((argmax2d_reg_kernel<1, true>), dim3(cdiv(rows, 4)), dim3(256), 0, st, hm, idx, xy, maxval, rows, W); break;
    default: hipLaunchKernelGGL(argmax2d_kernel, dim3(rows), dim3(256), 0, st, hm, idx, xy, maxval, H * W, W);
  }
  MI_CHECK_LAUNCH("argmax2d");
  return MI355_OK;
}
extern "C" int mi355_softargmax(const float* hm, float* uv, int rows, int H, int W, float beta, float out_scale, void* stream) {
  if (!hm || !uv || rows < 1 || H < 1 || W < 1) MI_FAIL(MI355_EINVAL, "softargmax: bad args");
  hipStream_t st = as_stream(stream);
  switch (row_form(H * W, hm)) {
    case 1: hipLaunchKernelGGL((softargmax_reg_kernel<4, false>), dim3(rows), dim3(256), 0, st, hm, uv, rows, W, beta, out_scale); break;
    case 2: hipLaunchKernelGGL((softargmax_reg_kernel<4, true>), dim3(cdiv(rows, 4)), dim3(256), 0, st, hm, uv, rows, W, beta, out_scale); break;
    case 3: hipLaunchKernelGGL((softargmax_reg_kernel<1, true>), dim3(cdiv(rows, 4)), dim3(256), 0, st, hm, uv, rows, W, beta, out_scale); break;
    default: hipLaunchKernelGGL(softargmax_kernel, dim3(rows), dim3(256), 0, st, hm, uv, H * W, W, beta, out_scale);
  }
  MI_CHECK_LAUNCH("softargmax");
  return MI355_OK;
}
extern "C" int mi355_kl_heatmap(const float* pred, const float* target, const float* weight, float eps, float* loss_rows,
                                float* unit_grad, int rows, int HW, float inv_count, void* stream) {
  if (!pred || !target || !loss_rows || rows < 1 || HW < 1) MI_FAIL(MI355_EINVAL, "kl_heatmap: bad args");
  hipStream_t st = as_stream(stream);
  switch (row_form(HW, pred, target, unit_grad)) {
    case 1: hipLaunchKernelGGL((kl_heatmap_reg_kernel<4, false>), dim3(rows), dim3(256), 0, st, pred, target, weight, eps, loss_rows, unit_grad, rows, inv_count); break;
    case 2: hipLaunchKernelGGL((kl_heatmap_reg_kernel<4, true>), dim3(cdiv(rows, 4)), dim3(256), 0, st, pred, target, weight, eps, loss_rows, unit_grad, rows, inv_count); break;
    case 3: hipLaunchKernelGGL((kl_heatmap_reg_kernel<1, true>), dim3(cdiv(rows, 4)), dim3(256), 0, st, pred, target, weight, eps, loss_rows, unit_grad, rows, inv_count); break;
    default: hipLaunchKernelGGL(kl_heatmap_kernel, dim3(rows), dim3(256), 0, st, pred, target, weight, eps, loss_rows, unit_grad, HW, inv_count);
  }
  MI_CHECK_LAUNCH("kl_heatmap");
  return MI355_OK;
}
extern "C" int mi355_reduce_sum(const float* in, float* out, int n, float scale, void* stream) {
  if (!in || !out || n < 1) MI_FAIL(MI355_EINVAL, "reduce_sum: bad args");
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, as_stream(stream), in, out, n, scale);
  MI_CHECK_LAUNCH("reduce_sum");
  return MI355_OK;
}
extern "C" int mi355_scale_by_dev(const float* in, const float* g_dev, float* out, long n, void* stream) {
  if (!in || !out || !g_dev || n < 1) MI_FAIL(MI355_EINVAL, "scale_by_dev: bad args");
  int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(scale_by_dev_kernel, dim3(grid), dim3(256), 0, as_stream(stream), in, g_dev, out, n);
  MI_CHECK_LAUNCH("scale_by_dev");
  return MI355_OK;
}
extern "C" int mi355_scale_feature(const void* in, const float* g_dev, void* out, long n, int dtype, void* stream) {
  const int per = dtype == MI355_BF16 ? 8 : 4;
  if (!in || !out || !g_dev || n < 1 || (dtype != MI355_BF16 && dtype != MI355_F32) || n % per)
    MI_FAIL(MI355_EINVAL, "scale_feature: bad args (n=%ld dtype=%d)", n, dtype);
  const long nvec = n / per;
  int grid = (int)((nvec + 255) / 256); if (grid > 8192) grid = 8192;
  if (dtype == MI355_BF16)
    hipLaunchKernelGGL(scale_feature_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t*)in, g_dev, (bf16_t*)out, nvec);
  else
    hipLaunchKernelGGL(scale_feature_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)in, g_dev, (float*)out, nvec);
  MI_CHECK_LAUNCH("scale_feature");
  return MI355_OK;
}
extern "C" int mi355_pseudo_label(const float* xy, const float* patch, int radius, int div, int S, int kind, const float* extra,
                                  int normalise, float* gt, float* gf, int B, int K, void* stream) {
  if (!xy || !patch || radius < 0 || radius > 6 || div < 1 || S < 1 || kind < 0 || kind > 2 || B < 1 || K < 1 || K > 64)
    MI_FAIL(MI355_EINVAL, "pseudo_label: bad args (radius=%d div=%d S=%d kind=%d B=%d K=%d)", radius, div, S, kind, B, K);
  const int form = row_form(S * S, extra, gt, gf);
  if (form == 1) hipLaunchKernelGGL(pseudo_label_reg_kernel<4>, dim3(B * K), dim3(256), 0, as_stream(stream), xy, patch, radius, div, S, kind, extra, normalise, gt, gf, K);
  else if (form == 2) hipLaunchKernelGGL(pseudo_label_reg_kernel<1>, dim3(B * K), dim3(256), 0, as_stream(stream), xy, patch, radius, div, S, kind, extra, normalise, gt, gf, K);
  else hipLaunchKernelGGL(pseudo_label_kernel, dim3(B * K), dim3(256), 0, as_stream(stream), xy, patch, radius, div, S, kind, extra, normalise, gt, gf, K);
  MI_CHECK_LAUNCH("pseudo_label");
  return MI355_OK;
}
extern "C" int mi355_bilinear_up(const float* in, float* out, int rows, int h, int w, int H, int W, float alpha, int accumulate, void* stream) {
  if (!in || !out || rows < 1 || h < 1 || w < 1 || H < 1 || W < 1) MI_FAIL(MI355_EINVAL, "bilinear_up: bad args");
  long total = (long)rows * H * W; int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(bilinear_up_kernel, dim3(grid), dim3(256), 0, as_stream(stream), in, out, rows, h, w, H, W, alpha, accumulate);
  MI_CHECK_LAUNCH("bilinear_up");
  return MI355_OK;
}
extern "C" int mi355_pck_dists(const float* pred_xy, const float* tgt_xy, float* dists, int rows, float norm_x, float norm_y, void* stream) {
  if (!pred_xy || !tgt_xy || !dists || rows < 1) MI_FAIL(MI355_EINVAL, "pck_dists: bad args");
  hipLaunchKernelGGL(pck_dists_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, as_stream(stream), pred_xy, tgt_xy, dists, rows, norm_x, norm_y);
  MI_CHECK_LAUNCH("pck_dists");
  return MI355_OK;
}
