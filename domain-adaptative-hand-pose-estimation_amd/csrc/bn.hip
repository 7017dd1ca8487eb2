// BatchNorm2d over NHWC rows (+ fused ReLU / residual add), train + eval forward and backward,
// and the bias-gradient column sum.  All of it is HBM-bound: 16-byte vector loads, per-thread
// channel ownership, Welford/Chan statistics in fp32, slice partials combined in a fixed order
// (bitwise reproducible; no atomics).  Slice partials stay slice-major: a channel-major layout makes the finalize
// kernels 35 % faster but the scattered 4-byte partial writes cost the statistics kernels twice that (measured).
#include "common.h"
#include "fp8_common.h"
#include <stdlib.h>
#include <string>

struct BnPlan { int cpr, TX, TY, colgroups, nslices, rows_per_slice; };

static BnPlan bn_plan(long rows, int C, int CH, long slice_cap_override = 0) {
  BnPlan p; p.cpr = C / CH;
  p.TX = p.cpr < 256 ? p.cpr : 256;
  // TX must divide 256
  while (256 % p.TX) --p.TX;
  p.TY = 256 / p.TX;
  p.colgroups = cdiv(p.cpr, p.TX);
  long want = rows / ((long)p.TY * 8); if (want < 1) want = 1;
  static const long slice_cap = getenv("MI355_BN_SLICES") ? atol(getenv("MI355_BN_SLICES")) : 1024;
  long cap = (slice_cap_override ? slice_cap_override : slice_cap) / p.colgroups; if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  long rps = (rows + want - 1) / want; rps = ((rps + p.TY - 1) / p.TY) * p.TY;
  p.rows_per_slice = (int)rps; p.nslices = (int)((rows + rps - 1) / rps);
  return p;
}

// the backward reduction (two rows per trip, 134 VGPRs, 3 blocks per CU) gets 768 slices = one resident wave of blocks
static long bwd_slices() { static const long v = getenv("MI355_BN_BWD_SLICES") ? atol(getenv("MI355_BN_BWD_SLICES")) : 768; return v; }
extern "C" size_t mi355_bn_workspace(long rows, int C) {
  int ns = 0;
  for (int ch = 4; ch <= 8; ch += 4) {     // fp32 / bf16 chunking, forward / backward plan: the largest
    BnPlan p = bn_plan(rows, C, ch), q = bn_plan(rows, C, ch, bwd_slices());
    if (p.nslices > ns) ns = p.nslices;
    if (q.nslices > ns) ns = q.nslices;
  }
  size_t n = (size_t)ns * C * 3;
  if (n < 32768) n = 32768;              // the resident backward's partials: <= 256 blocks x 64 channels x 2 sums
  return (n + 4 * (size_t)C) * sizeof(float);
}
extern "C" size_t mi355_colsum_workspace(long rows, int C) { return mi355_bn_workspace(rows, C); }

// -------------------------------------------------------------------------------- forward statistics
// partial[slice][c] = (n, mean, M2)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ partial, long rows,
                                                        int C, int TX, int rows_per_slice) {
  constexpr int CH = Chunk<T>::N;
  __shared__ float sh[256 * CH * 2 + 256];
  const int TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int chunk = blockIdx.x * TX + tx;
  const bool colok = chunk * CH < C;
  const long r0 = (long)blockIdx.y * rows_per_slice;
  long r1 = r0 + rows_per_slice; if (r1 > rows) r1 = rows;
  float mean[CH], m2[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { mean[e] = 0.f; m2[e] = 0.f; }
  float n = 0.f;
  if (colok)
    for (long r = r0 + ty; r < r1; r += TY) {
      float v[CH]; Chunk<T>::load(x + (size_t)r * C + (size_t)chunk * CH, v);
      n += 1.f; const float inv = 1.f / n;
#pragma unroll
      for (int e = 0; e < CH; ++e) { float d = v[e] - mean[e]; mean[e] += d * inv; m2[e] += d * (v[e] - mean[e]); }
    }
  // combine the TY row-lanes of each column (Chan et al.), fixed order
  float* smean = sh; float* sm2 = sh + 256 * CH; float* sn = sh + 512 * CH;
#pragma unroll
  for (int e = 0; e < CH; ++e) { smean[threadIdx.x * CH + e] = mean[e]; sm2[threadIdx.x * CH + e] = m2[e]; }
  sn[threadIdx.x] = n;
  __syncthreads();
  if (ty == 0 && colok) {
    for (int j = 1; j < TY; ++j) {
      const int o = j * TX + tx; const float nb = sn[o];
      if (nb > 0.f) {
        const float nt = n + nb, f = nb / nt;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          float d = smean[o * CH + e] - mean[e];
          mean[e] += d * f; m2[e] += sm2[o * CH + e] + d * d * n * f;
        }
        n = nt;
      }
    }
    float* out = partial + ((size_t)blockIdx.y * C + (size_t)chunk * CH) * 3;
#pragma unroll
    for (int e = 0; e < CH; ++e) { out[e * 3] = n; out[e * 3 + 1] = mean[e]; out[e * 3 + 2] = m2[e]; }
  }
}

// One wave per channel (4 channels per 256-thread block): lane l folds slices l, l+64, ... (<= 16 of them, all loads
// issued up front), then a shuffle-down tree folds the 64 lanes (lower lane = left operand, so the order is fixed and
// the result bitwise reproducible); lane 0 updates the running statistics and emits scale/shift.
__device__ __forceinline__ void chan_combine(float& n, float& mean, float& m2, float nb, float mb, float vb) {
  if (nb > 0.f) { const float nt = n + nb, f = nb / nt, d = mb - mean; mean += d * f; m2 += vb + d * d * n * f; n = nt; }
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nslices, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   int64_t* nbt, float* save_mean, float* save_invstd, float* scale_shift, float eps,
                                   float momentum, int repeats) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += repeats;
  if (c >= C) return;                      // wave-uniform
  // per-channel scalars first: their latency overlaps the partial loads instead of trailing the fold
  const float g = gamma[c], b = beta[c];
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int base = 0; base < nslices; base += 1024) {     // (conv-fused statistics can bring more than 1024 slices)
    float qn[16], qm[16], qv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int s = base + lane + 64 * j;
      if (s < nslices) { const float* q = partial + ((size_t)s * C + c) * 3; qn[j] = q[0]; qm[j] = q[1]; qv[j] = q[2]; }
      else { qn[j] = 0.f; qm[j] = 0.f; qv[j] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) chan_combine(n, mean, m2, qn[j], qm[j], qv[j]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_down(n, o, 64), mb = __shfl_down(mean, o, 64), vb = __shfl_down(m2, o, 64);
    chan_combine(n, mean, m2, nb, mb, vb);
  }
  if (lane != 0) return;
  const float var = m2 / n;
  const float invstd = 1.0f / sqrtf(var + eps);
  save_mean[c] = mean; save_invstd[c] = invstd;
  // `repeats` identical forward passes over the same batch (training step B -> C reuse) = that many momentum updates
  const float uvar = n > 1.f ? m2 / (n - 1.f) : var;
  for (int r = 0; r < repeats; ++r) {
    rm = (1.f - momentum) * rm + momentum * mean;
    rv = (1.f - momentum) * rv + momentum * uvar;
  }
  if (running_mean && repeats > 0) running_mean[c] = rm;
  if (running_var && repeats > 0) running_var[c] = rv;
  const float sc = g * invstd;
  scale_shift[c] = sc; scale_shift[C + c] = b - mean * sc;
}

// Many slices (statistics fused into a conv epilogue: one per 128-row tile, up to thousands): one 256-thread block per
// channel instead of one wave.  Thread t folds slices t, t+256, ..., then the shuffle tree per wave and the four waves in
// wave order -- again a fixed order.
__global__ __launch_bounds__(256) void bn_finalize_wide_kernel(const float* __restrict__ partial, int nslices, int C, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float* running_mean, float* running_var,
                                        int64_t* nbt, float* save_mean, float* save_invstd, float* scale_shift, float eps,
                                        float momentum, int repeats) {
  __shared__ float red[4][3];
  const int c = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (c == 0 && t == 0 && nbt) *nbt += repeats;
  const float g = gamma[c], b = beta[c];
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  float n = 0.f, mean = 0.f, m2 = 0.f;
  for (int base = 0; base < nslices; base += 2048) {
    float qn[8], qm[8], qv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = base + t + 256 * j;
      if (s < nslices) { const float* q = partial + ((size_t)s * C + c) * 3; qn[j] = q[0]; qm[j] = q[1]; qv[j] = q[2]; }
      else { qn[j] = 0.f; qm[j] = 0.f; qv[j] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) chan_combine(n, mean, m2, qn[j], qm[j], qv[j]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float nb = __shfl_down(n, o, 64), mb = __shfl_down(mean, o, 64), vb = __shfl_down(m2, o, 64);
    chan_combine(n, mean, m2, nb, mb, vb);
  }
  if (lane == 0) { red[wave][0] = n; red[wave][1] = mean; red[wave][2] = m2; }
  __syncthreads();
  if (t != 0) return;
  n = red[0][0]; mean = red[0][1]; m2 = red[0][2];
#pragma unroll
  for (int w = 1; w < 4; ++w) chan_combine(n, mean, m2, red[w][0], red[w][1], red[w][2]);
  const float var = m2 / n;
  const float invstd = 1.0f / sqrtf(var + eps);
  save_mean[c] = mean; save_invstd[c] = invstd;
  const float uvar = n > 1.f ? m2 / (n - 1.f) : var;
  for (int r = 0; r < repeats; ++r) {
    rm = (1.f - momentum) * rm + momentum * mean;
    rv = (1.f - momentum) * rv + momentum * uvar;
  }
  if (running_mean && repeats > 0) running_mean[c] = rm;
  if (running_var && repeats > 0) running_var[c] = rv;
  const float sc = g * invstd;
  scale_shift[c] = sc; scale_shift[C + c] = b - mean * sc;
}

// Q8 (bf16 only, 'fp8' compute mode): besides y, the e4m3 copy q8 = saturate(y * q8_state[0]) of the values as stored (the
// operand of the fp8 conv that consumes y: no stand-alone quantisation pass) and max |y| into q8_state[2] for the next scale.
template <typename T, bool EVAL, bool Q8 = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                        const float* __restrict__ scale_shift, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ rm,
                                                        const float* __restrict__ rv, float eps, long rows, int C, int TX, int relu,
                                                        unsigned char* __restrict__ mask, unsigned char* __restrict__ q8 = nullptr,
                                                        float* __restrict__ q8_state = nullptr) {
  constexpr int CH = Chunk<T>::N;
  static_assert(!Q8 || CH == 8, "fp8 side output: bf16 activations only");
  const int TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int chunk = blockIdx.x * TX + tx;
  const bool colok = chunk * CH < C;
  if (!colok && !Q8) return;                   // (Q8: every thread takes part in the block-wide amax at the end)
  const int cpr = C / CH;
  float sc[CH], sh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    const int c = chunk * CH + e;
    if (!colok) { sc[e] = 0.f; sh[e] = 0.f; }
    else if (EVAL) { sc[e] = gamma[c] / sqrtf(rv[c] + eps); sh[e] = beta[c] - rm[c] * sc[e]; }
    else { sc[e] = scale_shift[c]; sh[e] = scale_shift[C + c]; }
  }
  float qscale = 0.f, amax = 0.f; unsigned seen = 0;
  if constexpr (Q8) { qscale = q8_state[0]; seen = fp8_amax_seen(q8_state); }
  for (long r = (long)blockIdx.y * TY + ty; colok && r < rows; r += (long)gridDim.y * TY) {
    const size_t off = (size_t)r * C + (size_t)chunk * CH;
    float v[CH]; Chunk<T>::load(x + off, v);
#pragma unroll
    for (int e = 0; e < CH; ++e) v[e] = v[e] * sc[e] + sh[e];
    if (res) { float w[CH]; Chunk<T>::load(res + off, w);
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] += w[e]; }
    if (mask) {           // one bit per element: y > 0, what the backward's ReLU mask needs instead of re-reading y
      unsigned m = 0;
#pragma unroll
      for (int e = 0; e < CH; ++e) m |= (v[e] > 0.f ? 1u : 0u) << e;
      mask[(size_t)r * cpr + chunk] = (unsigned char)m;
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] = v[e] < 0.f ? 0.f : v[e]; }   // keeps NaN, like ATen relu
    Chunk<T>::store(y + off, v);
    if constexpr (Q8) {
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { w[e] = (float)(T)v[e]; amax = fmaxf(amax, fabsf(w[e])); }      // the stored (rounded) values
      // two neighbouring chunks of a row (lanes tx, tx+1: same row, always active together; C % 128 == 0) -> one 16-byte store
      const uint2 o = pack8_fp8<false>(w, qscale);
      const unsigned nx = __shfl_down(o.x, 1, 64), ny = __shfl_down(o.y, 1, 64);
      if ((tx & 1) == 0) *reinterpret_cast<uint4*>(q8 + off) = make_uint4(o.x, o.y, nx, ny);
    }
  }
  if constexpr (Q8) fp8_record_amax(amax, q8_state, seen);
}

// -------------------------------------------------------------------------------- backward
// partial[slice][c] = (sum dy_eff, sum dy_eff * xhat)
// RELU (compile time, so that every variant carries only the registers it needs: the bf16 apply kernel sits right at the
// 128-VGPR occupancy step): 0 none, 1 mask from y > 0, 2 mask recomputed from x, 3 mask from the forward's bit mask
template <typename T, bool XHAT, int RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ partial, long rows, int C, int TX, int rows_per_slice,
                                                             const unsigned char* __restrict__ mask) {
  constexpr int relu = RELU;
  constexpr int CH = Chunk<T>::N;
  __shared__ float sh[256 * CH * 2];
  const int cpr = C / CH;
  const int TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int chunk = blockIdx.x * TX + tx;
  const bool colok = chunk * CH < C;
  const long r0 = (long)blockIdx.y * rows_per_slice;
  long r1 = r0 + rows_per_slice; if (r1 > rows) r1 = rows;
  float s1[CH], s2[CH], mu[CH], is[CH], sc[CH], sft[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { s1[e] = 0.f; s2[e] = 0.f; mu[e] = 0.f; is[e] = 0.f; sc[e] = 0.f; sft[e] = 0.f; }
  if (colok && XHAT) {
#pragma unroll
    for (int e = 0; e < CH; ++e) { mu[e] = mean[chunk * CH + e]; is[e] = invstd[chunk * CH + e]; }
    if (relu == 2) {   // forward's exact scale/shift (bn_finalize_kernel): the ReLU mask is recomputed from x, y is not read
#pragma unroll
      for (int e = 0; e < CH; ++e) { sc[e] = gamma[chunk * CH + e] * is[e]; sft[e] = beta[chunk * CH + e] - mu[e] * sc[e]; }
    }
  }
  auto fold = [&](float (&g)[CH], const float (&v)[CH], const float (&o)[CH], unsigned mb) {
    if (relu == 1) {
#pragma unroll
      for (int e = 0; e < CH; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f; }
    else if (relu == 3) {
#pragma unroll
      for (int e = 0; e < CH; ++e) g[e] = ((mb >> e) & 1u) ? g[e] : 0.f; }
    if (XHAT) {
      if (relu == 2) {
#pragma unroll
        for (int e = 0; e < CH; ++e) g[e] = (v[e] * sc[e] + sft[e]) > 0.f ? g[e] : 0.f; }
#pragma unroll
      for (int e = 0; e < CH; ++e) { s1[e] += g[e]; s2[e] += g[e] * ((v[e] - mu[e]) * is[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < CH; ++e) s1[e] += g[e];
    }
  };
  if (colok) {
    long r = r0 + ty;
    for (; r + TY < r1; r += 2 * TY) {          // two rows per trip: twice the loads in flight
      const size_t off0 = (size_t)r * C + (size_t)chunk * CH, off1 = off0 + (size_t)TY * C;
      float g0[CH], v0[CH], o0[CH], g1[CH], v1[CH], o1[CH];
      Chunk<T>::load(dy + off0, g0); Chunk<T>::load(dy + off1, g1);
      if (XHAT) { Chunk<T>::load(x + off0, v0); Chunk<T>::load(x + off1, v1); }
      if (relu == 1) { Chunk<T>::load(y + off0, o0); Chunk<T>::load(y + off1, o1); }
      unsigned m0 = 0, m1 = 0;
      if (relu == 3) { m0 = mask[(size_t)r * cpr + chunk]; m1 = mask[(size_t)(r + TY) * cpr + chunk]; }
      fold(g0, v0, o0, m0); fold(g1, v1, o1, m1);
    }
    if (r < r1) {
      const size_t off = (size_t)r * C + (size_t)chunk * CH;
      float g[CH], v[CH], o[CH]; Chunk<T>::load(dy + off, g);
      if (XHAT) Chunk<T>::load(x + off, v);
      if (relu == 1) Chunk<T>::load(y + off, o);
      fold(g, v, o, relu == 3 ? (unsigned)mask[(size_t)r * cpr + chunk] : 0u);
    }
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) { sh[threadIdx.x * CH + e] = s1[e]; sh[256 * CH + threadIdx.x * CH + e] = s2[e]; }
  __syncthreads();
  if (ty == 0 && colok) {
    for (int j = 1; j < TY; ++j) {
      const int o = j * TX + tx;
#pragma unroll
      for (int e = 0; e < CH; ++e) { s1[e] += sh[o * CH + e]; s2[e] += sh[256 * CH + o * CH + e]; }
    }
    float* out = partial + ((size_t)blockIdx.y * C + (size_t)chunk * CH) * 2;
#pragma unroll
    for (int e = 0; e < CH; ++e) { out[e * 2] = s1[e]; out[e * 2 + 1] = s2[e]; }
  }
}

// coeff[c] = (k0 = gamma*invstd, k1 = mean(dy_eff), k2 = mean(dy_eff*xhat)); dgamma/dbeta (=|+=)
// one wave per channel, fixed-order shuffle tree (see bn_finalize_kernel)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nslices, int C, float inv_rows,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd, float* dgamma,
                                       float* dbeta, int accumulate, float* coeff) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const float g = coeff ? gamma[c] : 0.f, is = coeff ? invstd[c] : 0.f;       // scalars first (latency under the partial loads)
  const float db0 = (accumulate && dbeta) ? dbeta[c] : 0.f, dg0 = (accumulate && dgamma) ? dgamma[c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  for (int base = 0; base < nslices; base += 1024) {     // (conv-fused reductions can bring more than 1024 slices)
    float q1[16], q2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int s = base + lane + 64 * j;
      if (s < nslices) { const float* q = partial + ((size_t)s * C + c) * 2; q1[j] = q[0]; q2[j] = q[1]; } else { q1[j] = 0.f; q2[j] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) { s1 += q1[j]; s2 += q2[j]; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
  if (lane != 0) return;
  if (dbeta) dbeta[c] = db0 + s1;
  if (dgamma) dgamma[c] = dg0 + s2;
  if (coeff) { coeff[c] = g * is; coeff[C + c] = s1 * inv_rows; coeff[2 * C + c] = s2 * inv_rows; }
}

// Q8: also the e5m2 copy of dx (operand of the fp8 input-gradient GEMM of the conv in front of this BatchNorm) + its amax
template <typename T, int RELU, bool Q8 = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ coeff, const float* __restrict__ beta,
                                                            T* __restrict__ dx, T* __restrict__ dres,
                                                            long rows, int C, int TX, const unsigned char* __restrict__ mask,
                                                            unsigned char* __restrict__ q8 = nullptr, float* __restrict__ q8_state = nullptr) {
  constexpr int CH = Chunk<T>::N;
  constexpr int relu = RELU;
  static_assert(!Q8 || CH == 8, "fp8 side output: bf16 gradients only");
  const int TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int chunk = blockIdx.x * TX + tx;
  const bool colok = chunk * CH < C;
  if (!colok && !Q8) return;
  const int cpr = C / CH;
  float k0[CH], k1[CH], k2[CH], mu[CH], is[CH], sh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    const int c = colok ? chunk * CH + e : 0;
    k0[e] = coeff[c]; k1[e] = coeff[C + c]; k2[e] = coeff[2 * C + c]; mu[e] = mean[c]; is[e] = invstd[c];
    sh[e] = (relu == 2) ? beta[c] - mu[e] * k0[e] : 0.f;      // k0 = gamma*invstd = forward scale
  }
  float qscale = 0.f, amax = 0.f; unsigned seen = 0;
  if constexpr (Q8) { qscale = q8_state[0]; seen = fp8_amax_seen(q8_state); }
  auto finish = [&](size_t off, float (&g)[CH], float (&v)[CH], const float (&o)[CH], unsigned mb) {
    if (relu == 1) {
#pragma unroll
      for (int e = 0; e < CH; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f; }
    else if (relu == 3) {
#pragma unroll
      for (int e = 0; e < CH; ++e) g[e] = ((mb >> e) & 1u) ? g[e] : 0.f; }
    else if (relu == 2) {
#pragma unroll
      for (int e = 0; e < CH; ++e) g[e] = (v[e] * k0[e] + sh[e]) > 0.f ? g[e] : 0.f; }
    if (dres) Chunk<T>::store(dres + off, g);
#pragma unroll
    for (int e = 0; e < CH; ++e) v[e] = k0[e] * (g[e] - k1[e] - (v[e] - mu[e]) * is[e] * k2[e]);
    Chunk<T>::store(dx + off, v);
    if constexpr (Q8) {
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { w[e] = (float)(T)v[e]; amax = fmaxf(amax, fabsf(w[e])); }
      const uint2 o = pack8_fp8<true>(w, qscale);
      const unsigned nx = __shfl_down(o.x, 1, 64), ny = __shfl_down(o.y, 1, 64);
      if ((tx & 1) == 0) *reinterpret_cast<uint4*>(q8 + off) = make_uint4(o.x, o.y, nx, ny);
    }
  };
  // two rows per trip: twice the loads in flight per wave (the kernel is latency-, not issue-bound)
  const long stride = (long)gridDim.y * TY;
  long r = colok ? (long)blockIdx.y * TY + ty : rows;
  for (; r + stride < rows; r += 2 * stride) {
    const size_t off0 = (size_t)r * C + (size_t)chunk * CH, off1 = (size_t)(r + stride) * C + (size_t)chunk * CH;
    float g0[CH], v0[CH], o0[CH], g1[CH], v1[CH], o1[CH];
    Chunk<T>::load(dy + off0, g0); Chunk<T>::load(x + off0, v0);
    Chunk<T>::load(dy + off1, g1); Chunk<T>::load(x + off1, v1);
    if (relu == 1) { Chunk<T>::load(y + off0, o0); Chunk<T>::load(y + off1, o1); }
    unsigned m0 = 0, m1 = 0;
    if (relu == 3) { m0 = mask[(size_t)r * cpr + chunk]; m1 = mask[(size_t)(r + stride) * cpr + chunk]; }
    finish(off0, g0, v0, o0, m0); finish(off1, g1, v1, o1, m1);
  }
  if (r < rows) {
    const size_t off = (size_t)r * C + (size_t)chunk * CH;
    float g[CH], v[CH], o[CH]; Chunk<T>::load(dy + off, g); Chunk<T>::load(x + off, v);
    if (relu == 1) Chunk<T>::load(y + off, o);
    finish(off, g, v, o, relu == 3 ? (unsigned)mask[(size_t)r * cpr + chunk] : 0u);
  }
  if constexpr (Q8) fp8_record_amax(amax, q8_state, seen);
}

// -------------------------------------------------------------------------------- backward, tensor resident in LDS
// Small tensors (x and dy together within the chip's LDS: <= 16.8 MB each at 256 CUs x 160 KB) take ONE launch instead of
// reduce + finalize + apply: a block per CU holds its [rows / R][64 or 32 channels] tile of x and dy in LDS between the
// reduction and the apply pass, so both tensors are read from HBM once (3 tensor passes instead of 5) and two launch
// boundaries disappear.  The R blocks of a channel group exchange their partial sums through device memory inside the launch
// (guide, Guideline 16 counter form: write-through `sc1` partial stores drained by every storing wave, one agent-scope
// arrive add per block, one relaxed poller per block with a bounded spin, `sc1` loads of the partials; the sums are folded
// in a fixed order: bitwise reproducible).  All blocks must be resident together: the grid never exceeds the CU count.
struct BnResArgs {
  const void* dy; const void* x; const float* mean; const float* invstd; const float* gamma; const float* beta;
  void* dx; void* dres; float* dgamma; float* dbeta; const unsigned char* mask;
  float* partial;                       // [G][R][2 * channels per group]
  long rows; int C, G, R, rpb, keep, accumulate; float inv_rows;      // keep: rows of a block's tile that stay in LDS
  unsigned spin_limit;                  // grid-barrier poll iterations before a block gives up (and poisons its outputs)
};
// One 64-bit arrival counter per grid size, never reset: a launch of n blocks moves it from one multiple of n to the next, so
// a block that drew ticket v waits for the counter to reach (v / n + 1) * n (64 bits: no wrap within the life of a process;
// the 32-bit form would have wrapped after ~1e5 training iterations and broken every grid size that is not a power of two).
// Launches of this kernel on one device must not overlap in time (this library issues them on one stream;
// MI355_BN_RESIDENT=0 otherwise).  A block whose spin gives up (a co-resident kernel kept one of the grid's blocks off the chip
// for longer than the limit) FAILS LOUDLY: it counts the event in bn_res_err[0] (sticky until mi355_bn_resident_reset) and
// poisons everything it writes -- its dx rows, and dgamma / dbeta if it owns them -- with NaN instead of continuing on
// incomplete sums; the host side raises on the counter (mi355.ops.bn_resident_check: train1.py once per epoch, DAStep.check_health,
// bench.py, smoke).
#define BN_RES_MAXBLK 1024
#define BN_RES_KR 0        // tile rows per thread held in registers (8 VGPRs each)
#define BN_RES_NT 1024     // threads per block
__device__ unsigned long long bn_res_sync[BN_RES_MAXBLK + 1];     // [n] arrivals of the n-block launches
__device__ unsigned bn_res_err[4];                                // [0]: spins that gave up since load / reset

// returns false (in every thread of the block) when the spin gave up
__device__ __forceinline__ bool bn_res_grid_barrier(unsigned nblk, int t, unsigned spin_limit, int* ok_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its sc1 partial stores have left
  __syncthreads();
  if (t == 0) {
    int ok = 1;
    unsigned long long* st = bn_res_sync + nblk;
    const unsigned long long v = __hip_atomic_fetch_add(st, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (v / nblk + 1ull) * nblk;
    if (v + 1ull != target) {             // (the last arriver has nothing to wait for)
      unsigned spins = 0;
      while (__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > spin_limit) {       // default ~0.3 s: a block that never became resident
          __hip_atomic_fetch_add(bn_res_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    *ok_lds = ok;
  }
  __syncthreads();
  return *ok_lds != 0;
}

// A block's tile, by row: the first KR * NT / 8 rows live in REGISTERS (row ty + k NT / 8 in slot k of its thread), the
// next `keep` rows in LDS, the rest is streamed from HBM in both passes.  Built with KR = 0, NT = 1024: 512 threads with
// 8 register rows each (the register file holds more than the LDS) measured slower on every size but one -- two waves per
// SIMD hide less latency than the extra residency saves (64x64x64x64 bf16: 36.1 vs 31.0 us; 64x1024x16x16: 32.6 vs 28.5).
template <typename T, int RELU, int KR, int NT>
__global__ __launch_bounds__(NT) void bn_bwd_resident_kernel(BnResArgs p) {
  constexpr int CH = Chunk<T>::N, GC = 8 * CH, RL = NT / 8, NW = NT / 64, NL = NT / (GC * 2), RR = KR * RL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, tx = t & 7, ty = t >> 3, lane = t & 63, wave = t >> 6;
  const int g = blockIdx.x % p.G, r = blockIdx.x / p.G;
  const long row0 = (long)r * p.rpb;
  long row1 = row0 + p.rpb; if (row1 > p.rows) row1 = p.rows;
  const int nrows = row1 > row0 ? (int)(row1 - row0) : 0;
  const int keep = p.keep;
  uint4* xs = reinterpret_cast<uint4*>(smem);
  uint4* ds = xs + (size_t)keep * 8;
  float* red = reinterpret_cast<float*>(ds + (size_t)keep * 8);      // [NW][GC][2], later [NL][GC*2]
  float* tot = red + NW * GC * 2;                                    // [GC*2]
  int* bar_ok = reinterpret_cast<int*>(tot + GC * 2);                // the grid barrier's verdict, broadcast to the block
  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);
  const int C = p.C, cpr = C / CH, c0 = g * GC + tx * CH, chunk = c0 / CH;
  float mu[CH], is[CH], sc[CH], sft[CH], s1[CH], s2[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    mu[e] = p.mean[c0 + e]; is[e] = p.invstd[c0 + e]; s1[e] = 0.f; s2[e] = 0.f;
    sc[e] = p.gamma[c0 + e] * is[e];                                  // forward scale (bn_finalize_kernel)
    sft[e] = (RELU == 2) ? p.beta[c0 + e] - mu[e] * sc[e] : 0.f;
  }
  auto mask_of = [&](int lr) -> unsigned { return (RELU == 3) ? (unsigned)p.mask[(size_t)(row0 + lr) * cpr + chunk] : 0u; };
  auto fold = [&](const uint4& qxv, const uint4& qdv, unsigned mb) {
    float v[CH], gq[CH]; Chunk<T>::unpack(qxv, v); Chunk<T>::unpack(qdv, gq);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      if (RELU == 2) gq[e] = (v[e] * sc[e] + sft[e]) > 0.f ? gq[e] : 0.f;
      if (RELU == 3) gq[e] = ((mb >> e) & 1u) ? gq[e] : 0.f;
      s1[e] += gq[e]; s2[e] += gq[e] * ((v[e] - mu[e]) * is[e]);
    }
  };
  // ---- pass 1: HBM -> registers / LDS, per-thread sums over its rows
  uint4 rx[KR > 0 ? KR : 1], rd[KR > 0 ? KR : 1];
  {
    unsigned mk[KR > 0 ? KR : 1];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int lr = ty + k * RL;
      rx[k] = make_uint4(0, 0, 0, 0); rd[k] = make_uint4(0, 0, 0, 0); mk[k] = 0;
      if (lr < nrows) {
        const size_t off = (size_t)(row0 + lr) * C + c0;
        rx[k] = *reinterpret_cast<const uint4*>(X + off); rd[k] = *reinterpret_cast<const uint4*>(DY + off); mk[k] = mask_of(lr);
      }
    }
#pragma unroll
    for (int k = 0; k < KR; ++k)
      if (ty + k * RL < nrows) fold(rx[k], rd[k], mk[k]);
  }
  for (int base = RR + ty; base < nrows; base += 4 * RL) {          // four rows in flight per thread
    uint4 qx[4], qd[4]; unsigned mk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lr = base + k * RL;
      if (lr < nrows) {
        const size_t off = (size_t)(row0 + lr) * C + c0;
        qx[k] = *reinterpret_cast<const uint4*>(X + off); qd[k] = *reinterpret_cast<const uint4*>(DY + off); mk[k] = mask_of(lr);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lr = base + k * RL, li = lr - RR;
      if (lr < nrows) {
        if (li < keep) { xs[li * 8 + tx] = qx[k]; ds[li * 8 + tx] = qd[k]; }
        fold(qx[k], qd[k], mk[k]);
      }
    }
  }
  // fold the row lanes: the 8 of a wave by xor shuffles, the NW waves in wave order through LDS
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < CH; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
  }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < CH; ++e) { red[(wave * GC + tx * CH + e) * 2] = s1[e]; red[(wave * GC + tx * CH + e) * 2 + 1] = s2[e]; }
  }
  __syncthreads();
  float* part = p.partial + ((size_t)g * p.R) * (GC * 2);
  if (t < GC * 2) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) a += red[w * GC * 2 + t];
    __hip_atomic_store(part + (size_t)r * (GC * 2) + t, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: write-through
  }
  const bool exchanged = bn_res_grid_barrier((unsigned)(p.G * p.R), t, p.spin_limit, bar_ok);
  // ---- the group's totals: NL lanes per value over the R blocks (fixed order), then the lanes in order
  {
    // sc1 buffer loads (aux 16), eight in flight per thread: atomic loads would be waited for one at a time
    const int i = t % (GC * 2), s = t / (GC * 2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(part, 0, (unsigned)(p.R * GC * 2 * 4), 0x00020000);
    float a = 0.f;
    for (int rr0 = s; rr0 < p.R; rr0 += 8 * NL) {
      float q[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int rr = rr0 + k * NL;
        q[k] = rr < p.R ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (rr * (GC * 2) + i) * 4, 0, 16)) : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) a += q[k];
    }
    red[s * GC * 2 + i] = a;
  }
  __syncthreads();
  if (t < GC * 2) {
    float a = 0.f;
#pragma unroll
    for (int s = 0; s < NL; ++s) a += red[s * GC * 2 + t];
    tot[t] = a;
  }
  __syncthreads();
  float k1[CH], k2[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    s1[e] = tot[(tx * CH + e) * 2]; s2[e] = tot[(tx * CH + e) * 2 + 1];
    if (!exchanged) { s1[e] = __builtin_nanf(""); s2[e] = __builtin_nanf(""); }      // incomplete sums must not pass for a gradient
    k1[e] = s1[e] * p.inv_rows; k2[e] = s2[e] * p.inv_rows;
  }
  if (r == 0 && ty == 0) {
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      if (p.dbeta) p.dbeta[c0 + e] = (p.accumulate ? p.dbeta[c0 + e] : 0.f) + s1[e];
      if (p.dgamma) p.dgamma[c0 + e] = (p.accumulate ? p.dgamma[c0 + e] : 0.f) + s2[e];
    }
  }
  // ---- pass 2: register rows, LDS rows, then the rest of the tile streamed from HBM again -> dx (and the masked dy for
  // the residual branch)
  T* __restrict__ DX = reinterpret_cast<T*>(p.dx);
  T* __restrict__ DR = reinterpret_cast<T*>(p.dres);
  auto finish = [&](int lr, const uint4& qxv, const uint4& qdv, unsigned mb) {
    const size_t off = (size_t)(row0 + lr) * C + c0;
    float v[CH], gq[CH]; Chunk<T>::unpack(qxv, v); Chunk<T>::unpack(qdv, gq);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      if (RELU == 2) gq[e] = (v[e] * sc[e] + sft[e]) > 0.f ? gq[e] : 0.f;
      if (RELU == 3) gq[e] = ((mb >> e) & 1u) ? gq[e] : 0.f;
    }
    if (DR) Chunk<T>::store(DR + off, gq);
#pragma unroll
    for (int e = 0; e < CH; ++e) v[e] = sc[e] * (gq[e] - k1[e] - (v[e] - mu[e]) * is[e] * k2[e]);
    Chunk<T>::store(DX + off, v);
  };
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int lr = ty + k * RL;
    if (lr < nrows) finish(lr, rx[k], rd[k], mask_of(lr));
  }
  const int nlds = nrows - RR < keep ? nrows - RR : keep;
  for (int li = ty; li < nlds; li += RL) finish(RR + li, xs[li * 8 + tx], ds[li * 8 + tx], mask_of(RR + li));
  for (int base = RR + keep + ty; base < nrows; base += 4 * RL) {
    uint4 qx[4], qd[4]; unsigned mk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lr = base + k * RL;
      if (lr < nrows) {
        const size_t off = (size_t)(row0 + lr) * C + c0;
        qx[k] = *reinterpret_cast<const uint4*>(X + off); qd[k] = *reinterpret_cast<const uint4*>(DY + off); mk[k] = mask_of(lr);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lr = base + k * RL;
      if (lr < nrows) finish(lr, qx[k], qd[k], mk[k]);
    }
  }
}

// g <- bit of mask set ? g : 0, in place (the stand-alone form of the masking the consumers of a residual block's fork
// gradient normally do on the fly: mi355_conv_dgrad_masked_acc, mi355_bn_bwd with relu_mask)
template <typename T>
__global__ __launch_bounds__(256) void apply_relu_mask_kernel(T* __restrict__ g, const unsigned char* __restrict__ mask, size_t nchunks) {
  constexpr int CH = Chunk<T>::N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * 256) {
    float v[CH]; Chunk<T>::load(g + i * CH, v);
    const unsigned mb = mask[i];
#pragma unroll
    for (int e = 0; e < CH; ++e) v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
    Chunk<T>::store(g + i * CH, v);
  }
}

// Stem: BatchNorm + ReLU + MaxPool2d(3, 2, 1) in one pass (resnet.py:27-28 of the torchvision stem).  The normalised
// 64-channel 128x128 map (134 MB at B=64) is never written: every window element is normalised on the fly (rounded to T as the
// stand-alone apply pass would store it, so maxima and arg-max bytes are the same bit for bit), the backward recomputes the
// ReLU mask from the conv output as for any BatchNorm without a residual.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const T* __restrict__ x, const float* __restrict__ scale_shift, T* __restrict__ y,
                                                               uint8_t* __restrict__ arg, int N, int H, int W, int C, int Ho, int Wo) {
  constexpr int CH = Chunk<T>::N;
  const int cpr = C / CH;
  const unsigned total = (unsigned)N * Ho * Wo * cpr;            // (< 2^31: checked by the host; 32-bit index arithmetic -- 64-bit divisions cost more than the loads)
  for (unsigned id = blockIdx.x * 256u + threadIdx.x; id < total; id += gridDim.x * 256u) {
    const int ch = (int)(id % (unsigned)cpr); unsigned r = id / (unsigned)cpr;
    const int ox = (int)(r % (unsigned)Wo); r /= (unsigned)Wo; const int oy = (int)(r % (unsigned)Ho); const int n = (int)(r / (unsigned)Ho);
    float sc[CH], sh[CH], best[CH]; int bi[CH]; bool first = true;
#pragma unroll
    for (int e = 0; e < CH; ++e) { sc[e] = scale_shift[ch * CH + e]; sh[e] = scale_shift[C + ch * CH + e]; best[e] = -INFINITY; bi[e] = 0; }
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy - 1 + kh; if (iy < 0 || iy >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = 2 * ox - 1 + kw; if (ix < 0 || ix >= W) continue;
        float v[CH]; Chunk<T>::load(x + (((size_t)n * H + iy) * W + ix) * C + (size_t)ch * CH, v);
        const int code = kh * 3 + kw;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          float u = v[e] * sc[e] + sh[e];
          u = u < 0.f ? 0.f : u;                           // keeps NaN
          u = (float)(T)u;                                 // the value the apply pass would have stored
          if (first) bi[e] = code;
          if (u > best[e] || u != u) { best[e] = u; bi[e] = code; }
        }
        first = false;
      }
    }
    const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + (size_t)ch * CH;
    Chunk<T>::store(y + o, best);
    store_bytes<CH>(arg + o, bi);
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nslices, int C, float* out, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float q[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { const int i = lane + 64 * j; q[j] = (i < nslices) ? partial[((size_t)i * C + c) * 2] : 0.f; }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += q[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) out[c] = (accumulate ? out[c] : 0.f) + s;
}

// -------------------------------------------------------------------------------- host
static int q8_check(const void* q8_out, const float* q8_state, int dtype, int C) {
  if (!q8_out && !q8_state) return 0;
  if (!q8_out || !q8_state || dtype != MI355_BF16 || C % 16) MI_FAIL(MI355_EINVAL, "bn: the fp8 side output needs both q8 pointers, bf16 activations and C %% 16 == 0");
  return 0;
}
static int bn_check(long rows, int C, int dtype, int* CH) {
  if (dtype != MI355_F32 && dtype != MI355_BF16) MI_FAIL(MI355_EINVAL, "bn: bad dtype");
  *CH = dtype == MI355_BF16 ? 8 : 4;
  if (rows < 1 || C < 1 || C % *CH) MI_FAIL(MI355_EINVAL, "bn: C=%d must be a positive multiple of %d (rows=%ld)", C, *CH, rows);
  return 0;
}
// grid = what is resident at once (256 CUs x blocks per CU at the kernel's register count): a second, partial wave of blocks
// costs more than longer grid-stride loops (backward apply, two rows per trip, 122 VGPRs -> 4 blocks per CU)
static dim3 apply_grid(const BnPlan& p, long rows, bool backward = false) {
  long gy = rows / ((long)p.TY * 4); if (gy < 1) gy = 1;
  static const long fwd_cap = getenv("MI355_BN_APPLY_CAP") ? atol(getenv("MI355_BN_APPLY_CAP")) : 2048;
  static const long bwd_cap = getenv("MI355_BN_BWD_CAP") ? atol(getenv("MI355_BN_BWD_CAP")) : 1024;
  long cap = (backward ? bwd_cap : fwd_cap) / p.colgroups; if (cap < 1) cap = 1;
  if (gy > cap) gy = cap;
  return dim3(p.colgroups, (unsigned)gy);
}

extern "C" int mi355_bn_train_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, int64_t* nbt, float* save_mean, float* save_invstd,
                                  long rows, int C, float eps, float momentum, int stat_updates, int relu, int dtype, void* ws,
                                  size_t ws_bytes, void* relu_mask, void* q8_out, float* q8_state, void* stream) {
  int CH; if (int e = bn_check(rows, C, dtype, &CH)) return e;
  if (int e = q8_check(q8_out, q8_state, dtype, C)) return e;
  unsigned char* mk = reinterpret_cast<unsigned char*>(relu_mask);
  if (stat_updates < 0 || stat_updates > 8) MI_FAIL(MI355_EINVAL, "bn_train_fwd: stat_updates=%d", stat_updates);
  if (!ws || ws_bytes < mi355_bn_workspace(rows, C)) MI_FAIL(MI355_EWORKSPACE, "bn_train_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  BnPlan p = bn_plan(rows, C, CH);
  float* partial = reinterpret_cast<float*>(ws);
  float* ss = partial + (size_t)p.nslices * C * 3;
  dim3 g(p.colgroups, p.nslices);
  const double nb = (double)rows * C * (dtype == MI355_BF16 ? 2.0 : 4.0);
  char lab[96];
  if (prof_on()) snprintf(lab, sizeof(lab), "rows%ld C%d%s%s", rows, C, relu ? " relu" : "", residual ? " +res" : "");
  {
    ProfScope ps(st, 0.0, nb, 1, prof_on() ? (std::string("bn_stats ") + lab).c_str() : nullptr);
    if (dtype == MI355_BF16) hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)x, partial, rows, C, p.TX, p.rows_per_slice);
    else hipLaunchKernelGGL(bn_stats_kernel<float>, g, dim3(256), 0, st, (const float*)x, partial, rows, C, p.TX, p.rows_per_slice);
  }
  {
    ProfScope ps(st, 0.0, 12.0 * p.nslices * C, 1, prof_on() ? (std::string("bn_finalize ") + lab).c_str() : nullptr);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, partial, p.nslices, C, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, ss, eps, momentum, stat_updates);
  }
  dim3 ga = apply_grid(p, rows);
  const float* nf = nullptr;
  ProfScope ps(st, 0.0, nb * (residual ? 3 : 2), 1, prof_on() ? (std::string("bn_apply ") + lab).c_str() : nullptr);
  if (q8_out) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false, true>), ga, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, (const float*)ss, nf, nf, nf, nf, eps, rows, C, p.TX, relu, mk, (unsigned char*)q8_out, q8_state);
  else if (dtype == MI355_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false>), ga, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, (const float*)ss, nf, nf, nf, nf, eps, rows, C, p.TX, relu, mk);
  else hipLaunchKernelGGL((bn_apply_kernel<float, false>), ga, dim3(256), 0, st, (const float*)x, (const float*)residual, (float*)y, (const float*)ss, nf, nf, nf, nf, eps, rows, C, p.TX, relu, mk);
  MI_CHECK_LAUNCH("bn_train_fwd");
  return MI355_OK;
}

// Same as mi355_bn_train_fwd, but the per-channel statistics partials [nslices][C][n, mean, M2] were already produced
// by the convolution that wrote x (mi355_conv_fwd_stats / mi355_conv_dgrad_stats): no statistics pass over x.
extern "C" int mi355_bn_train_fwd_partials(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, int64_t* nbt, float* save_mean,
                                           float* save_invstd, long rows, int C, float eps, float momentum, int stat_updates,
                                           int relu, int dtype, const float* partial, int nslices, float* scale_shift,
                                           void* relu_mask, void* q8_out, float* q8_state, void* stream) {
  int CH; if (int e = bn_check(rows, C, dtype, &CH)) return e;
  if (int e = q8_check(q8_out, q8_state, dtype, C)) return e;
  unsigned char* mk = reinterpret_cast<unsigned char*>(relu_mask);
  if (stat_updates < 0 || stat_updates > 8) MI_FAIL(MI355_EINVAL, "bn_train_fwd_partials: stat_updates=%d", stat_updates);
  if (!partial || nslices < 1 || !scale_shift) MI_FAIL(MI355_EINVAL, "bn_train_fwd_partials: partial / scale_shift missing");
  hipStream_t st = as_stream(stream);
  BnPlan p = bn_plan(rows, C, CH);
  static const int wide_min = getenv("MI355_BN_WIDE_FINALIZE") ? atoi(getenv("MI355_BN_WIDE_FINALIZE")) : 512;
  const double nb = (double)rows * C * (dtype == MI355_BF16 ? 2.0 : 4.0);
  char lab[96];
  if (prof_on()) snprintf(lab, sizeof(lab), "rows%ld C%d ns%d%s%s", rows, C, nslices, relu ? " relu" : "", residual ? " +res" : "");
  {
    ProfScope ps(st, 0.0, 12.0 * nslices * C, 1, prof_on() ? (std::string("bn_finalize ") + lab).c_str() : nullptr);
    if (nslices >= wide_min) hipLaunchKernelGGL(bn_finalize_wide_kernel, dim3(C), dim3(256), 0, st, partial, nslices, C, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, scale_shift, eps, momentum, stat_updates);
    else hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, partial, nslices, C, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, scale_shift, eps, momentum, stat_updates);
  }
  dim3 ga = apply_grid(p, rows);
  const float* nf = nullptr;
  ProfScope ps(st, 0.0, nb * (residual ? 3 : 2), 1, prof_on() ? (std::string("bn_apply ") + lab).c_str() : nullptr);
  if (q8_out) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false, true>), ga, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, (const float*)scale_shift, nf, nf, nf, nf, eps, rows, C, p.TX, relu, mk, (unsigned char*)q8_out, q8_state);
  else if (dtype == MI355_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false>), ga, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, (const float*)scale_shift, nf, nf, nf, nf, eps, rows, C, p.TX, relu, mk);
  else hipLaunchKernelGGL((bn_apply_kernel<float, false>), ga, dim3(256), 0, st, (const float*)x, (const float*)residual, (float*)y, (const float*)scale_shift, nf, nf, nf, nf, eps, rows, C, p.TX, relu, mk);
  MI_CHECK_LAUNCH("bn_train_fwd_partials");
  return MI355_OK;
}

// mi355_bn_train_fwd_partials (statistics partials from the conv's epilogue) + ReLU + MaxPool2d(3, 2, 1): x [N][H][W][C] ->
// y_pool [N][Ho][Wo][C] and the window-position bytes mi355_maxpool_bwd takes; the normalised map is not written.
extern "C" int mi355_bn_relu_maxpool_fwd_partials(const void* x, void* y_pool, uint8_t* argidx, const float* gamma, const float* beta,
                                                  float* running_mean, float* running_var, int64_t* nbt, float* save_mean,
                                                  float* save_invstd, int N, int H, int W, int C, float eps, float momentum,
                                                  int stat_updates, int dtype, const float* partial, int nslices, float* scale_shift,
                                                  void* stream) {
  const long rows = (long)N * H * W;
  int CH; if (int e = bn_check(rows, C, dtype, &CH)) return e;
  if (stat_updates < 0 || stat_updates > 8) MI_FAIL(MI355_EINVAL, "bn_relu_maxpool_fwd_partials: stat_updates=%d", stat_updates);
  if (!partial || nslices < 1 || !scale_shift || !argidx || !y_pool) MI_FAIL(MI355_EINVAL, "bn_relu_maxpool_fwd_partials: null argument");
  if (rows * (C / CH) >= (1L << 31)) MI_FAIL(MI355_EINVAL, "bn_relu_maxpool_fwd_partials: %ld chunks exceed the kernel's 32-bit index: split the batch", rows * (C / CH));
  hipStream_t st = as_stream(stream);
  static const int wide_min = getenv("MI355_BN_WIDE_FINALIZE") ? atoi(getenv("MI355_BN_WIDE_FINALIZE")) : 512;
  char lab[96];
  if (prof_on()) snprintf(lab, sizeof(lab), "rows%ld C%d ns%d", rows, C, nslices);
  {
    ProfScope ps(st, 0.0, 12.0 * nslices * C, 1, prof_on() ? (std::string("bn_finalize ") + lab).c_str() : nullptr);
    if (nslices >= wide_min) hipLaunchKernelGGL(bn_finalize_wide_kernel, dim3(C), dim3(256), 0, st, partial, nslices, C, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, scale_shift, eps, momentum, stat_updates);
    else hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, partial, nslices, C, gamma, beta, running_mean, running_var, nbt, save_mean, save_invstd, scale_shift, eps, momentum, stat_updates);
  }
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / CH);
  int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
  ProfScope ps(st, 0.0, (double)rows * C * (dtype == MI355_BF16 ? 2.0 : 4.0) * 1.25, 1, prof_on() ? (std::string("bn_relu_maxpool ") + lab).c_str() : nullptr);
  if (dtype == MI355_BF16) hipLaunchKernelGGL(bn_relu_maxpool_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const float*)scale_shift, (bf16_t*)y_pool, argidx, N, H, W, C, Ho, Wo);
  else hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (const float*)scale_shift, (float*)y_pool, argidx, N, H, W, C, Ho, Wo);
  MI_CHECK_LAUNCH("bn_relu_maxpool_fwd_partials");
  return MI355_OK;
}

extern "C" int mi355_bn_eval_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                                 const float* running_mean, const float* running_var, long rows, int C, float eps, int relu,
                                 int dtype, void* stream) {
  int CH; if (int e = bn_check(rows, C, dtype, &CH)) return e;
  hipStream_t st = as_stream(stream);
  BnPlan p = bn_plan(rows, C, CH);
  dim3 ga = apply_grid(p, rows);
  const float* nf = nullptr;
  if (dtype == MI355_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, true>), ga, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, nf, gamma, beta, running_mean, running_var, eps, rows, C, p.TX, relu, (unsigned char*)nullptr);
  else hipLaunchKernelGGL((bn_apply_kernel<float, true>), ga, dim3(256), 0, st, (const float*)x, (const float*)residual, (float*)y, nf, gamma, beta, running_mean, running_var, eps, rows, C, p.TX, relu, (unsigned char*)nullptr);
  MI_CHECK_LAUNCH("bn_eval_fwd");
  return MI355_OK;
}

// ---- resident backward: plan + launch (returns false when the tensor does not fit / the mode is not covered)
struct BnResPlan { int G, R, rpb, keep; size_t lds, max_lds; };
static int g_bn_resident = -1;       // run-time switch (mi355_bn_set_resident); -1: the environment decides
// The one-launch backward needs every block resident at once.  A caller that runs other kernels BESIDE the backward on the same
// device (a collective overlapped with it: mi355/da_step.py) switches it off for that stretch: a CU held by the other kernel
// would keep a block out while the resident ones spin.  Returns the previous setting.
extern "C" int mi355_bn_set_resident(int on) {
  const int prev = g_bn_resident;
  g_bn_resident = on < 0 ? -1 : (on ? 1 : 0);
  return prev;
}
static bool bn_resident_plan(long rows, int C, int CH, BnResPlan* q) {
  static const bool env_on = !(getenv("MI355_BN_RESIDENT") && atoi(getenv("MI355_BN_RESIDENT")) == 0);
  if (g_bn_resident == 0 || (g_bn_resident < 0 && !env_on)) return false;
  static int ncu = 0; static size_t max_lds = 0;
  if (!ncu) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return false;
    max_lds = (size_t)v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    ncu = v > 256 ? 256 : v;      // mi355_bn_workspace reserves partials for at most 256 blocks x 64 channels x 2 sums
  }
  const int GC = 8 * CH;
  if (C % GC) return false;
  static const long min_bytes = getenv("MI355_BN_RESIDENT_MIN") ? atol(getenv("MI355_BN_RESIDENT_MIN")) : 0;
  if (rows * C * (16 / CH) < min_bytes) return false;       // (small tensors: three short launches beat the in-launch exchange)
  q->G = C / GC;
  if (q->G > ncu) return false;
  long R = ncu / q->G; if (R > rows) R = rows;
  q->rpb = (int)((rows + R - 1) / R);
  q->R = (int)((rows + q->rpb - 1) / q->rpb);
  static const long max_bytes = getenv("MI355_BN_RESIDENT_MAX") ? atol(getenv("MI355_BN_RESIDENT_MAX")) : (1L << 62);
  if (rows * C * (16 / CH) > max_bytes) return false;
  const size_t scratch = (size_t)(BN_RES_NT / 64 * GC * 2 + GC * 2 + 4) * sizeof(float);      // sums, totals, barrier verdict
  long keep = (long)((max_lds - scratch) / 256);        // tile rows (128 B of x + 128 B of dy each) that fit beside the scratch
  const long reg_rows = (long)BN_RES_KR * (BN_RES_NT / 8);
  const long rest = q->rpb > reg_rows ? q->rpb - reg_rows : 0;
  if (keep > rest) keep = rest;
  q->keep = (int)keep;
  q->lds = (size_t)keep * 256 + scratch;
  q->max_lds = max_lds;
  return (size_t)q->G * q->R <= (size_t)ncu && q->G * q->R <= BN_RES_MAXBLK;
}
static unsigned g_bn_res_spin = 1u << 19;      // grid-barrier poll iterations (~0.3 s) before a block gives up
// Test hook: shrink (or restore, 0 = default) the spin bound so that a give-up can be provoked on purpose.
extern "C" int mi355_bn_resident_set_spin_limit(unsigned limit) {
  g_bn_res_spin = limit ? limit : (1u << 19);
  return MI355_OK;
}
#define BN_RES_NOT_RESIDENT 1      // (internal) the kernel cannot hold one block per CU: take the three-launch form
template <typename T, int RELU>
static int bn_resident_launch(const BnResArgs& a, size_t lds, size_t max_lds, hipStream_t st) {
  static int fits = -1;      // does one block of this variant fit a CU at the largest LDS size the plan may ask for?
  auto kern = bn_bwd_resident_kernel<T, RELU, BN_RES_KR, BN_RES_NT>;
  if (fits < 0) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds) != hipSuccess)
      MI_FAIL(MI355_ELAUNCH, "bn_bwd: cannot raise the dynamic LDS limit to %zu bytes", max_lds);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, BN_RES_NT, max_lds) != hipSuccess) nb = 0;
    fits = nb >= 1 ? 1 : 0;
  }
  if (!fits) return BN_RES_NOT_RESIDENT;
  hipLaunchKernelGGL(kern, dim3(a.G * a.R), dim3(BN_RES_NT), lds, st, a);
  return MI355_OK;
}
// Number of grid-barrier spins that gave up since the library was loaded or mi355_bn_resident_reset (0 unless a block of a
// one-launch backward was kept off the chip).  Every such launch has poisoned its outputs with NaN.  Synchronises the device.
extern "C" int mi355_bn_resident_timeouts(unsigned* out) {
  unsigned v[4] = {0, 0, 0, 0};
  if (!out) MI_FAIL(MI355_EINVAL, "bn_resident_timeouts: out is null");
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(bn_res_err), sizeof(v)) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "bn_resident_timeouts: copy failed");
  *out = v[0];
  return MI355_OK;
}
// Clears the give-up count and the arrival counters (after a give-up has been reported and handled).  Synchronises the device.
extern "C" int mi355_bn_resident_reset(void) {
  static const unsigned long long zeros[BN_RES_MAXBLK + 1] = {0};
  static const unsigned zerr[4] = {0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) MI_FAIL(MI355_ELAUNCH, "bn_resident_reset: device synchronize failed");
  if (hipMemcpyToSymbol(HIP_SYMBOL(bn_res_sync), zeros, sizeof(zeros)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(bn_res_err), zerr, sizeof(zerr)) != hipSuccess) MI_FAIL(MI355_ELAUNCH, "bn_resident_reset: copy failed");
  return MI355_OK;
}

static void launch_bwd_apply(int dtype, int relu, dim3 ga, hipStream_t st, const void* dy, const void* x, const void* y, const float* save_mean,
                             const float* save_invstd, const float* coeff, const float* beta, void* dx, void* dresidual, long rows, int C,
                             int TX, const unsigned char* mk, void* q8_out, float* q8_state) {
#define MI_APP(T, R) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R>), ga, dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)y, save_mean, save_invstd, coeff, beta, (T*)dx, (T*)dresidual, rows, C, TX, mk)
#define MI_APQ(R) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, R, true>), ga, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)y, save_mean, save_invstd, coeff, beta, (bf16_t*)dx, (bf16_t*)dresidual, rows, C, TX, mk, (unsigned char*)q8_out, q8_state)
#define MI_APP4(T) do { if (relu == 0) MI_APP(T, 0); else if (relu == 1) MI_APP(T, 1); else if (relu == 2) MI_APP(T, 2); else MI_APP(T, 3); } while (0)
  if (q8_out) { if (relu == 0) MI_APQ(0); else if (relu == 1) MI_APQ(1); else if (relu == 2) MI_APQ(2); else MI_APQ(3); }
  else if (dtype == MI355_BF16) MI_APP4(bf16_t); else MI_APP4(float);
#undef MI_APP4
#undef MI_APQ
#undef MI_APP
}

extern "C" int mi355_bn_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_invstd, void* dx, void* dresidual, float* dgamma, float* dbeta, int accumulate,
                            long rows, int C, int relu, int dtype, void* ws, size_t ws_bytes, const void* relu_mask, void* q8_out,
                            float* q8_state, void* stream) {
  int CH; if (int e = bn_check(rows, C, dtype, &CH)) return e;
  if (!ws || ws_bytes < mi355_bn_workspace(rows, C)) MI_FAIL(MI355_EWORKSPACE, "bn_bwd: workspace too small");
  // relu: the mask comes from the bit mask the forward wrote (1 bit / element), else from y when it is given; without
  // either it is recomputed from x (only valid when the forward had no residual add): one tensor read less per pass
  const unsigned char* mk = reinterpret_cast<const unsigned char*>(relu_mask);
  if (relu) relu = mk ? 3 : (y ? 1 : 2);
  if (relu == 2 && !beta) MI_FAIL(MI355_EINVAL, "bn_bwd: relu without y needs beta");
  hipStream_t st = as_stream(stream);
  BnResPlan rp;
  if (relu != 1 && !q8_out && bn_resident_plan(rows, C, CH, &rp)) {
    if (ws_bytes < (size_t)rp.G * rp.R * (8 * CH) * 2 * sizeof(float)) MI_FAIL(MI355_EWORKSPACE, "bn_bwd: workspace too small for the one-launch partials");
    BnResArgs a;
    a.dy = dy; a.x = x; a.mean = save_mean; a.invstd = save_invstd; a.gamma = gamma; a.beta = beta; a.dx = dx; a.dres = dresidual;
    a.dgamma = dgamma; a.dbeta = dbeta; a.mask = mk; a.partial = reinterpret_cast<float*>(ws); a.rows = rows; a.C = C;
    a.G = rp.G; a.R = rp.R; a.rpb = rp.rpb; a.keep = rp.keep; a.accumulate = accumulate; a.inv_rows = 1.0f / (float)rows;
    a.spin_limit = g_bn_res_spin;
    int e;
    char lab[112];
    if (prof_on()) snprintf(lab, sizeof(lab), "bn_bwd_res rows%ld C%d relu%d%s keep%d/%d", rows, C, relu, dresidual ? " +dres" : "", rp.keep, rp.rpb);
    ProfScope ps(st, 0.0, (double)rows * C * (dtype == MI355_BF16 ? 2.0 : 4.0) * (dresidual ? 4 : 3), 1, prof_on() ? lab : nullptr);
    if (dtype == MI355_BF16) e = relu == 0 ? bn_resident_launch<bf16_t, 0>(a, rp.lds, rp.max_lds, st) : relu == 2 ? bn_resident_launch<bf16_t, 2>(a, rp.lds, rp.max_lds, st) : bn_resident_launch<bf16_t, 3>(a, rp.lds, rp.max_lds, st);
    else e = relu == 0 ? bn_resident_launch<float, 0>(a, rp.lds, rp.max_lds, st) : relu == 2 ? bn_resident_launch<float, 2>(a, rp.lds, rp.max_lds, st) : bn_resident_launch<float, 3>(a, rp.lds, rp.max_lds, st);
    if (e == MI355_OK) { MI_CHECK_LAUNCH("bn_bwd (resident)"); return MI355_OK; }
    if (e != BN_RES_NOT_RESIDENT) return e;
  }
  BnPlan p = bn_plan(rows, C, CH, bwd_slices());
  float* partial = reinterpret_cast<float*>(ws);
  float* coeff = partial + (size_t)p.nslices * C * 3;   // 3*C floats (4*C reserved)
  dim3 g(p.colgroups, p.nslices);
#define MI_RED(T, R) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true, R>), g, dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)y, save_mean, save_invstd, gamma, beta, partial, rows, C, p.TX, p.rows_per_slice, mk)
#define MI_RED4(T) do { if (relu == 0) MI_RED(T, 0); else if (relu == 1) MI_RED(T, 1); else if (relu == 2) MI_RED(T, 2); else MI_RED(T, 3); } while (0)
  const double nb = (double)rows * C * (dtype == MI355_BF16 ? 2.0 : 4.0);
  char lab[96];
  if (prof_on()) snprintf(lab, sizeof(lab), "rows%ld C%d relu%d%s", rows, C, relu, dresidual ? " +dres" : "");
  {
    ProfScope ps(st, 0.0, nb * 2, 1, prof_on() ? (std::string("bn_bwd_reduce ") + lab).c_str() : nullptr);
    if (dtype == MI355_BF16) MI_RED4(bf16_t); else MI_RED4(float);
  }
#undef MI_RED4
#undef MI_RED
  {
    ProfScope ps(st, 0.0, 12.0 * p.nslices * C, 1, prof_on() ? (std::string("bn_bwd_finalize ") + lab).c_str() : nullptr);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, partial, p.nslices, C, 1.0f / (float)rows, gamma, save_invstd, dgamma, dbeta, accumulate, coeff);
  }
  dim3 ga = apply_grid(p, rows, true);
  if (int e = q8_check(q8_out, q8_state, dtype, C)) return e;
  ProfScope ps(st, 0.0, nb * (dresidual ? 4 : 3), 1, prof_on() ? (std::string("bn_bwd_apply ") + lab).c_str() : nullptr);
  launch_bwd_apply(dtype, relu, ga, st, dy, x, y, save_mean, save_invstd, coeff, beta, dx, dresidual, rows, C, p.TX, mk, q8_out, q8_state);
  MI_CHECK_LAUNCH("bn_bwd");
  return MI355_OK;
}

// mi355_bn_bwd without its reduction pass: (sum dy_eff, sum dy_eff*xhat) partials [nslices][C][2] came out of the epilogue
// of the GEMM that produced dy (mi355_conv_dgrad_bnbwd / mi355_conv_fwd_bnbwd).  coeff: 3*C floats of scratch.
extern "C" int mi355_bn_bwd_partials(const void* dy, const void* x, const void* y, const float* gamma, const float* beta,
                                     const float* save_mean, const float* save_invstd, void* dx, void* dresidual, float* dgamma,
                                     float* dbeta, int accumulate, long rows, int C, int relu, int dtype, const float* partial,
                                     int nslices, float* coeff, const void* relu_mask, void* q8_out, float* q8_state, void* stream) {
  int CH; if (int e = bn_check(rows, C, dtype, &CH)) return e;
  if (!partial || nslices < 1 || !coeff) MI_FAIL(MI355_EINVAL, "bn_bwd_partials: partial / coeff missing");
  const unsigned char* mk = reinterpret_cast<const unsigned char*>(relu_mask);
  if (relu) relu = mk ? 3 : (y ? 1 : 2);
  if (relu == 2 && !beta) MI_FAIL(MI355_EINVAL, "bn_bwd_partials: relu without y needs beta");
  hipStream_t st = as_stream(stream);
  BnPlan p = bn_plan(rows, C, CH);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, partial, nslices, C, 1.0f / (float)rows, gamma, save_invstd, dgamma, dbeta, accumulate, coeff);
  dim3 ga = apply_grid(p, rows, true);
  if (int e = q8_check(q8_out, q8_state, dtype, C)) return e;
  launch_bwd_apply(dtype, relu, ga, st, dy, x, y, save_mean, save_invstd, coeff, beta, dx, dresidual, rows, C, p.TX, mk, q8_out, q8_state);
  MI_CHECK_LAUNCH("bn_bwd_partials");
  return MI355_OK;
}

extern "C" int mi355_apply_relu_mask(void* g, const void* relu_mask, long rows, int C, int dtype, void* stream) {
  int CH; if (int e = bn_check(rows, C, dtype, &CH)) return e;
  if (!g || !relu_mask) MI_FAIL(MI355_EINVAL, "apply_relu_mask: null pointer");
  const size_t nchunks = (size_t)rows * (C / CH);
  long grid = (long)((nchunks + 255) / 256); if (grid > 4096) grid = 4096;
  if (dtype == MI355_BF16) hipLaunchKernelGGL(apply_relu_mask_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), (bf16_t*)g, (const unsigned char*)relu_mask, nchunks);
  else hipLaunchKernelGGL(apply_relu_mask_kernel<float>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), (float*)g, (const unsigned char*)relu_mask, nchunks);
  MI_CHECK_LAUNCH("apply_relu_mask");
  return MI355_OK;
}

extern "C" int mi355_colsum(const void* dy, float* out, long rows, int C, int dtype, int accumulate, void* ws, size_t ws_bytes,
                            void* stream) {
  int CH; if (int e = bn_check(rows, C, dtype, &CH)) return e;
  if (!ws || ws_bytes < mi355_colsum_workspace(rows, C)) MI_FAIL(MI355_EWORKSPACE, "colsum: workspace too small");
  hipStream_t st = as_stream(stream);
  BnPlan p = bn_plan(rows, C, CH);
  float* partial = reinterpret_cast<float*>(ws);
  dim3 g(p.colgroups, p.nslices);
  if (dtype == MI355_BF16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, false, 0>), g, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, partial, rows, C, p.TX, p.rows_per_slice, (const unsigned char*)nullptr);
  else hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, false, 0>), g, dim3(256), 0, st, (const float*)dy, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, partial, rows, C, p.TX, p.rows_per_slice, (const unsigned char*)nullptr);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, partial, p.nslices, C, out, accumulate);
  MI_CHECK_LAUNCH("colsum");
  return MI355_OK;
}
