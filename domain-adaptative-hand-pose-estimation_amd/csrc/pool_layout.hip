// Stem max-pool 3x3 s2 p1 (NHWC, argmax kept as a window-position byte) and the NCHW<->NHWC edge
// conversions.  HBM-bound; 16-byte chunks per lane.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ arg,
                                                           int N, int H, int W, int C, int Ho, int Wo) {
  constexpr int CH = Chunk<T>::N;
  const int cpr = C / CH;
  const unsigned total = (unsigned)N * Ho * Wo * cpr;            // (< 2^31: checked by the host; 32-bit index arithmetic -- 64-bit divisions cost more than the loads)
  for (unsigned id = blockIdx.x * 256u + threadIdx.x; id < total; id += gridDim.x * 256u) {
    const int ch = (int)(id % (unsigned)cpr); unsigned r = id / (unsigned)cpr;
    const int ox = (int)(r % (unsigned)Wo); r /= (unsigned)Wo; const int oy = (int)(r % (unsigned)Ho); const int n = (int)(r / (unsigned)Ho);
    float best[CH]; int bi[CH]; bool first = true;
#pragma unroll
    for (int e = 0; e < CH; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy - 1 + kh; if (iy < 0 || iy >= H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = 2 * ox - 1 + kw; if (ix < 0 || ix >= W) continue;
        float v[CH]; Chunk<T>::load(x + (((size_t)n * H + iy) * W + ix) * C + (size_t)ch * CH, v);
        const int code = kh * 3 + kw;
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          if (first) bi[e] = code;                       // ATen: index starts at the first valid position
          if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = code; }
        }
        first = false;
      }
    }
    const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + (size_t)ch * CH;
    Chunk<T>::store(y + o, best);
    store_bytes<CH>(arg + o, bi);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ arg, T* __restrict__ dx,
                                                           int N, int H, int W, int C, int Ho, int Wo) {
  constexpr int CH = Chunk<T>::N;
  const int cpr = C / CH;
  const unsigned total = (unsigned)N * H * W * cpr;              // (< 2^31: checked by the host)
  for (unsigned id = blockIdx.x * 256u + threadIdx.x; id < total; id += gridDim.x * 256u) {
    const int ch = (int)(id % (unsigned)cpr); unsigned r = id / (unsigned)cpr;
    const int ix = (int)(r % (unsigned)W); r /= (unsigned)W; const int iy = (int)(r % (unsigned)H); const int n = (int)(r / (unsigned)H);
    float g[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) g[e] = 0.f;
    // Input row iy lies in window row oy0 = iy / 2 (at kernel row kh0 = iy - 2 oy0 + 1 = 1 or 2) and, when iy is odd, also in
    // oy0 + 1 (kh = 0); columns alike: at most 2 x 2 windows, visited as straight-line predicated code (the 3 x 3 loop with its
    // parity tests diverged inside every wave: 78 us for 185 MB)
    const int oy0 = iy >> 1, ox0 = ix >> 1;
    const int khs[2] = {iy - 2 * oy0 + 1, 0}, kws[2] = {ix - 2 * ox0 + 1, 0};
    const bool vy[2] = {oy0 < Ho, (iy & 1) && oy0 + 1 < Ho}, vx[2] = {ox0 < Wo, (ix & 1) && ox0 + 1 < Wo};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (vy[a] && vx[b]) {
          const size_t o = (((size_t)n * Ho + oy0 + a) * Wo + ox0 + b) * C + (size_t)ch * CH;
          float d[CH]; Chunk<T>::load(dy + o, d);
          unsigned ab[CH]; load_bytes<CH>(arg + o, ab);
          const unsigned code = (unsigned)(khs[a] * 3 + kws[b]);
#pragma unroll
          for (int e = 0; e < CH; ++e) if (ab[e] == code) g[e] += d[e];
        }
      }
    Chunk<T>::store(dx + (((size_t)n * H + iy) * W + ix) * C + (size_t)ch * CH, g);
  }
}

extern "C" int mi355_maxpool_fwd(const void* x, void* y, uint8_t* argidx, int N, int H, int W, int C, int dtype, void* stream) {
  const int CH = dtype == MI355_BF16 ? 8 : 4;
  if (C % CH || N < 1 || !argidx) MI_FAIL(MI355_EINVAL, "maxpool_fwd: bad args (C=%d)", C);
  if ((long)N * H * W * (C / CH) >= (1L << 31)) MI_FAIL(MI355_EINVAL, "maxpool_fwd: %ld chunks exceed the kernel's 32-bit index: split the batch", (long)N * H * W * (C / CH));
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  long total = (long)N * Ho * Wo * (C / CH);
  int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
  if (dtype == MI355_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, argidx, N, H, W, C, Ho, Wo);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, argidx, N, H, W, C, Ho, Wo);
  MI_CHECK_LAUNCH("maxpool_fwd");
  return MI355_OK;
}

extern "C" int mi355_maxpool_bwd(const void* dy, const uint8_t* argidx, void* dx, int N, int H, int W, int C, int dtype, void* stream) {
  const int CH = dtype == MI355_BF16 ? 8 : 4;
  if (C % CH || N < 1 || !argidx) MI_FAIL(MI355_EINVAL, "maxpool_bwd: bad args (C=%d)", C);
  if ((long)N * H * W * (C / CH) >= (1L << 31)) MI_FAIL(MI355_EINVAL, "maxpool_bwd: %ld chunks exceed the kernel's 32-bit index: split the batch", (long)N * H * W * (C / CH));
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  long total = (long)N * H * W * (C / CH);
  int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
  if (dtype == MI355_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t*)dy, argidx, (bf16_t*)dx, N, H, W, C, Ho, Wo);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)dy, argidx, (float*)dx, N, H, W, C, Ho, Wo);
  MI_CHECK_LAUNCH("maxpool_bwd");
  return MI355_OK;
}

// ---------------------------------------------------------------- NCHW fp32 -> NHWC T (zero-padded channels)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int HW, int Cpad) {
  constexpr int CH = Chunk<T>::N;
  const int cpr = Cpad / CH;
  const long total = (long)N * HW * cpr;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    // pixel fastest so the strided NCHW reads coalesce across lanes
    const int p = (int)(id % HW); long r = id / HW; const int ch = (int)(r % cpr); const int n = (int)(r / cpr);
    float v[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { const int c = ch * CH + e; v[e] = c < C ? x[((size_t)n * C + c) * HW + p] : 0.f; }
    Chunk<T>::store(y + ((size_t)n * HW + p) * Cpad + (size_t)ch * CH, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int C, int HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (p < HW && c < C) ? Elem<T>::ld(x + ((size_t)n * HW + p) * C + c) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (p < HW && c < C) y[((size_t)n * C + c) * HW + p] = tile[tx][r];
  }
}

extern "C" int mi355_nchw_to_nhwc(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype, void* stream) {
  const int CH = dtype == MI355_BF16 ? 8 : 4;
  if (Cpad % CH || Cpad < C) MI_FAIL(MI355_EINVAL, "nchw_to_nhwc: Cpad=%d must be a multiple of %d and >= C=%d", Cpad, CH, C);
  long total = (long)N * H * W * (Cpad / CH);
  int grid = (int)((total + 255) / 256); if (grid > 16384) grid = 16384;
  if (dtype == MI355_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), x, (bf16_t*)y, N, C, H * W, Cpad);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), x, (float*)y, N, C, H * W, Cpad);
  MI_CHECK_LAUNCH("nchw_to_nhwc");
  return MI355_OK;
}

extern "C" int mi355_nhwc_to_nchw(const void* x, float* y, int N, int C, int H, int W, int dtype, void* stream) {
  dim3 grid(cdiv(H * W, 32), cdiv(C, 32), N);
  if (dtype == MI355_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)x, y, N, C, H * W);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, y, N, C, H * W);
  MI_CHECK_LAUNCH("nhwc_to_nchw");
  return MI355_OK;
}

// ---------------------------------------------------------------- stem as a 4x4 conv over the space-to-depth image
// A 7x7 / stride-2 / pad-3 conv over a 3-channel image equals a 4x4 / unit-stride conv over the image folded 2x2 into channels:
// y[oy][ox] = sum over block rows by = oy-2 .. oy+1 (bx alike) and the 2x2 pixels (dy, dx) inside a block, with image row
// 2*by + dy = 2*oy - 3 + kh, i.e. kh = 2*(by - oy + 2) + dy - 1 (the combination kh = -1 carries a zero weight).  K = 16 taps x 16
// channels = 256 (four full K tiles) instead of 49 taps x 8 padded channels = 392 (seven, the last one nearly empty), and the folded
// image is half the bytes of the channel-padded one.  Folded channel = (dy*2 + dx)*4 + c, c = 3 is zero.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_s2d_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int H, int W) {
  const int H2 = H >> 1, W2 = W >> 1;
  const long total = (long)N * H2 * W2;
  const size_t plane = (size_t)H * W;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int bx = (int)(id % W2); long r = id / W2; const int by = (int)(r % H2); const int n = (int)(r / H2);
    float v[16];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float2 q = *reinterpret_cast<const float2*>(x + ((size_t)n * 3 + c) * plane + (size_t)(2 * by + dy) * W + 2 * bx);
        v[(dy * 2 + 0) * 4 + c] = q.x; v[(dy * 2 + 1) * 4 + c] = q.y;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q * 4 + 3] = 0.f;
    T* o = y + (size_t)id * 16;
    constexpr int CH = Chunk<T>::N;
#pragma unroll
    for (int k = 0; k < 16 / CH; ++k) {
      float g[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) g[e] = v[k * CH + e];
      Chunk<T>::store(o + k * CH, g);
    }
  }
}

extern "C" int mi355_nchw_to_s2d(const float* x, void* y, int N, int H, int W, int dtype, void* stream) {
  if (!x || !y || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) MI_FAIL(MI355_EINVAL, "nchw_to_s2d: a 3-channel image with even extents (got %dx%d)", H, W);
  if (dtype != MI355_BF16 && dtype != MI355_F32) MI_FAIL(MI355_EINVAL, "nchw_to_s2d: dtype %d", dtype);
  long total = (long)N * (H / 2) * (W / 2);
  int grid = (int)((total + 255) / 256); if (grid > 16384) grid = 16384;
  if (dtype == MI355_BF16) hipLaunchKernelGGL(nchw_to_s2d_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), x, (bf16_t*)y, N, H, W);
  else hipLaunchKernelGGL(nchw_to_s2d_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), x, (float*)y, N, H, W);
  MI_CHECK_LAUNCH("nchw_to_s2d");
  return MI355_OK;
}

// w fp32 [Co][7][7][3] (conv-form master) -> out `T` [Co][4][4][16]: the forward operand of the folded stem
template <typename T>
__global__ __launch_bounds__(256) void stem_s2d_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int Co) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= Co * 256) return;
  const int ch = id & 15, tap = (id >> 4) & 15, o = id >> 8;
  const int c = ch & 3, dy = ch >> 3, dx = (ch >> 2) & 1;
  const int kh = 2 * (tap >> 2) + dy - 1, kw = 2 * (tap & 3) + dx - 1;
  const float v = (c < 3 && kh >= 0 && kh < 7 && kw >= 0 && kw < 7) ? w[((o * 7 + kh) * 7 + kw) * 3 + c] : 0.f;
  Elem<T>::st(out + id, v);
}
// gs fp32 [Co][4][4][16] (weight gradient of the folded form) -> g fp32 [Co][7][7][3] (=, or += when accumulate)
__global__ __launch_bounds__(256) void stem_s2d_unpack_kernel(const float* __restrict__ gs, float* __restrict__ g, int Co, int accumulate) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= Co * 147) return;
  const int c = id % 3, kw = (id / 3) % 7, kh = (id / 21) % 7, o = id / 147;
  const int th = (kh + 1) >> 1, dy = (kh + 1) & 1, tw = (kw + 1) >> 1, dx = (kw + 1) & 1;
  const float v = gs[(o * 16 + th * 4 + tw) * 16 + (dy * 2 + dx) * 4 + c];
  g[id] = accumulate ? g[id] + v : v;
}

extern "C" int mi355_stem_s2d_pack(const float* w, void* out, int Co, int dtype, void* stream) {
  if (!w || !out || Co < 1) MI_FAIL(MI355_EINVAL, "stem_s2d_pack: bad args");
  if (dtype != MI355_BF16 && dtype != MI355_F32) MI_FAIL(MI355_EINVAL, "stem_s2d_pack: dtype %d", dtype);
  if (dtype == MI355_BF16) hipLaunchKernelGGL(stem_s2d_pack_kernel<bf16_t>, dim3(Co), dim3(256), 0, as_stream(stream), w, (bf16_t*)out, Co);
  else hipLaunchKernelGGL(stem_s2d_pack_kernel<float>, dim3(Co), dim3(256), 0, as_stream(stream), w, (float*)out, Co);
  MI_CHECK_LAUNCH("stem_s2d_pack");
  return MI355_OK;
}
extern "C" int mi355_stem_s2d_unpack_grad(const float* gs, float* g, int Co, int accumulate, void* stream) {
  if (!gs || !g || Co < 1) MI_FAIL(MI355_EINVAL, "stem_s2d_unpack_grad: bad args");
  hipLaunchKernelGGL(stem_s2d_unpack_kernel, dim3(cdiv((long)Co * 147, 256)), dim3(256), 0, as_stream(stream), gs, g, Co, accumulate);
  MI_CHECK_LAUNCH("stem_s2d_unpack_grad");
  return MI355_OK;
}
