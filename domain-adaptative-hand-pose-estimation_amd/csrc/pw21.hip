// 1x1 convolutions touching the K=21 heat-map tensors.  Heat-maps are NCHW fp32 (rows of H*W, what the
// loss kernels read), features NHWC.  K=21 is far below an MFMA tile, and every one of these ops moves
// >= 134 MB of feature map per launch at B=64 for <= 2.8 GFLOP, so they are HBM-bound VALU kernels:
// coalesced 16-byte feature loads, weights and heat-map slices staged in LDS, fp32 math.
#include "common.h"

#define PW_MAXK 32
#define PW_PIX 64   // pixels per block

// y[n][k][p] = bias[k] + sum_c x[n*HW+p][c] * w[k][c]
template <typename T>
__global__ __launch_bounds__(256) void pw_c2k_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ y, int HW, int C, int K, int wtr) {
  constexpr int CH = Chunk<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xstride = C * (int)sizeof(T) + 16;             // padded row: conflict-free per-lane row reads
  char* xs = smem;                                          // [PW_PIX][C] (+pad)
  float* ws = reinterpret_cast<float*>(smem + PW_PIX * xstride);   // [K][C]
  const int n = blockIdx.y, p0 = blockIdx.x * PW_PIX, t = threadIdx.x;
  const int cpr = C / CH;
  for (int id = t; id < PW_PIX * cpr; id += 256) {
    const int r = id / cpr, c = id % cpr;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p0 + r < HW) v = *reinterpret_cast<const uint4*>(x + ((size_t)n * HW + p0 + r) * C + (size_t)c * CH);
    *reinterpret_cast<uint4*>(xs + r * xstride + c * 16) = v;
  }
  for (int id = t; id < K * C; id += 256) ws[id] = wtr ? w[(size_t)(id % C) * K + id / C] : w[id];
  __syncthreads();
  const int pl = t & 63, kg = t >> 6;                       // lane = pixel, wave = k-group (wave-uniform weights)
  float acc[(PW_MAXK + 3) / 4];
#pragma unroll
  for (int j = 0; j < (PW_MAXK + 3) / 4; ++j) acc[j] = 0.f;
  for (int c = 0; c < cpr; ++c) {
    float v[CH]; Chunk<T>::load(reinterpret_cast<const T*>(xs + pl * xstride + c * 16), v);
#pragma unroll
    for (int j = 0; j < (PW_MAXK + 3) / 4; ++j) {
      const int k = kg + 4 * j;
      if (k < K) {
        const float* wk = ws + k * C + c * CH;
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[j] = fmaf(v[e], wk[e], acc[j]);
      }
    }
  }
  if (p0 + pl < HW) {
#pragma unroll
    for (int j = 0; j < (PW_MAXK + 3) / 4; ++j) {
      const int k = kg + 4 * j;
      if (k < K) y[((size_t)n * K + k) * HW + p0 + pl] = acc[j] + (bias ? bias[k] : 0.f);
    }
  }
}

// out[n*HW+p][c] = (bias[c] + sum_k y[n][k][p] * w[c][k] + residual) * scale
// Persistent over the 64-pixel tiles of all images: a block loads the weights of its channel group into LDS once (the K-strided
// gather of the [C][K] layout costs 42 cache lines per wave load; done per tile it was the larger part of the kernel's time) and
// then walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...; tile number = n * tiles_per_image + tile_in_image, which is also the
// statistics slice.
template <typename T>
__global__ __launch_bounds__(256) void pw_k2c_kernel(const float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
                                                      const T* __restrict__ res, const float* __restrict__ scale_dev, T* __restrict__ out,
                                                      int HW, int C, int K, int wtr, float* __restrict__ stat_partial, int tiles_per_img,
                                                      int ntiles) {
  constexpr int CH = Chunk<T>::N;
  constexpr int CG = 256 / 8;                               // 32 channel-chunks per block pass, 8 pixel groups
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ys = reinterpret_cast<float*>(smem);               // [K][PW_PIX]
  float* ws = ys + K * PW_PIX;                              // [K][CH/4][CG][4]  (transposed weights of this channel group)
  float* sp = ws + K * CG * CH;                             // [4 waves][CG*CH][3]   (statistics launches only)
  const int t = threadIdx.x;
  const int cbase = blockIdx.y * CG * CH;
  // weights of this channel group in LDS as [k][CH / 4][CG][4]: a thread's CH channels are CH / 4 float4 reads whose lanes sit
  // 16 bytes apart (conflict-free)
  constexpr int NSUB = CH / 4;
  for (int id = t; id < K * CG * CH; id += 256) {
    const int k = id / (CG * CH), c = id % (CG * CH);
    const int g = c / CH, e = c % CH;
    ws[((k * NSUB + (e >> 2)) * CG + g) * 4 + (e & 3)] =
        (cbase + c < C) ? (wtr ? w[(size_t)k * C + cbase + c] : w[(size_t)(cbase + c) * K + k]) : 0.f;
  }
  const int cg = t & (CG - 1), pg = t >> 5;                 // pg: 8 pixels each
  const int c0 = cbase + cg * CH;
  const bool cok = c0 < C;
  const float scale = scale_dev ? *scale_dev : 1.f;
  float b[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) b[e] = (bias && cok) ? bias[c0 + e] : 0.f;
  const int lane = t & 63, wave = t >> 6;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / tiles_per_img, p0 = (tile - n * tiles_per_img) * PW_PIX;
    __syncthreads();                                          // the previous tile's reads of ys (and of sp) are done
    for (int id = t; id < K * PW_PIX; id += 256) {
      const int k = id / PW_PIX, p = id % PW_PIX;
      ys[id] = (p0 + p < HW) ? y[((size_t)n * K + k) * HW + p0 + p] : 0.f;
    }
    __syncthreads();
    float sn = 0.f, smean[CH], sm2[CH];       // BatchNorm statistics of the stored (rounded) values, as in the gather epilogue
#pragma unroll
    for (int e = 0; e < CH; ++e) { smean[e] = 0.f; sm2[e] = 0.f; }
    if (cok) {
      float acc[8][CH];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[i][e] = 0.f;
      for (int k = 0; k < K; ++k) {
        float wv[CH];
#pragma unroll
        for (int q = 0; q < NSUB; ++q) {
          const float4 w4 = *reinterpret_cast<const float4*>(ws + ((k * NSUB + q) * CG + cg) * 4);
          wv[4 * q] = w4.x; wv[4 * q + 1] = w4.y; wv[4 * q + 2] = w4.z; wv[4 * q + 3] = w4.w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float yv = ys[k * PW_PIX + pg * 8 + i];
#pragma unroll
          for (int e = 0; e < CH; ++e) acc[i][e] = fmaf(yv, wv[e], acc[i][e]);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = p0 + pg * 8 + i;
        if (p >= HW) continue;
        const size_t off = ((size_t)n * HW + p) * C + c0;
        float v[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] = acc[i][e] + b[e];
        if (res) { float r[CH]; Chunk<T>::load(res + off, r);
#pragma unroll
          for (int e = 0; e < CH; ++e) v[e] += r[e]; }
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] *= scale;
        Chunk<T>::store(out + off, v);
        if (stat_partial) {
          sn += 1.f; const float inv = 1.f / sn;
#pragma unroll
          for (int e = 0; e < CH; ++e) { const float q = (float)(T)v[e]; const float d = q - smean[e]; smean[e] += d * inv; sm2[e] += d * (q - smean[e]); }
        }
      }
    }
    if (!stat_partial) continue;
    // fold the 8 pixel groups: lanes 32..63 onto 0..31 inside a wave, then the four waves in order through LDS (fixed order)
    {
      const float nb = __shfl_down(sn, 32, 64);
      const float nt = sn + nb, f = nt > 0.f ? nb / nt : 0.f;
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const float mb = __shfl_down(smean[e], 32, 64), vb = __shfl_down(sm2[e], 32, 64);
        const float d = mb - smean[e];
        smean[e] += d * f; sm2[e] += vb + d * d * sn * f;
      }
      sn = nt;
    }
    if (lane < 32) {
#pragma unroll
      for (int e = 0; e < CH; ++e) { float* q = sp + ((size_t)wave * CG * CH + cg * CH + e) * 3; q[0] = sn; q[1] = smean[e]; q[2] = sm2[e]; }
    }
    __syncthreads();
    if (t < CG * CH && cbase + t < C) {
      float nn = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) {
        const float* q = sp + ((size_t)w2 * CG * CH + t) * 3;
        const float nb = q[0];
        if (nb > 0.f) { const float nt = nn + nb, f = nb / nt, d = q[1] - mean; mean += d * f; m2 += q[2] + d * d * nn * f; nn = nt; }
      }
      float* o = stat_partial + ((size_t)tile * C + cbase + t) * 3;
      o[0] = nn; o[1] = mean; o[2] = m2;
    }
  }
}

// partial[slice][k][c] = sum over the slice's pixels of y[n][k][p] * x[n*HW+p][c]
template <typename T>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ y, float* __restrict__ partial,
                                                        int N, int HW, int C, int K, int pix_per_slice) {
  __shared__ __attribute__((aligned(16))) float ys[PW_MAXK * 64];   // [K][64 pixels]
  const int t = threadIdx.x, c = blockIdx.y * 256 + t;
  const long P = (long)N * HW;
  const long q0 = (long)blockIdx.x * pix_per_slice;
  long q1 = q0 + pix_per_slice; if (q1 > P) q1 = P;
  float acc[PW_MAXK];
#pragma unroll
  for (int k = 0; k < PW_MAXK; ++k) acc[k] = 0.f;
  for (long q = q0; q < q1; q += 64) {          // HW % 64 == 0 is required so a 64-pixel group never straddles images
    const int n = (int)(q / HW), p = (int)(q % HW);
    __syncthreads();
    for (int id = t; id < K * 64; id += 256) { const int k = id >> 6, j = id & 63; ys[id] = (q + j < q1) ? y[((size_t)n * K + k) * HW + p + j] : 0.f; }
    __syncthreads();
    if (c < C) {
      for (int j = 0; j < 64; j += 4) {
        float xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = (q + j + u < q1) ? Elem<T>::ld(x + (size_t)(q + j + u) * C + c) : 0.f;
#pragma unroll
        for (int k = 0; k < PW_MAXK; ++k) {
          if (k < K) {
            const float4 yv = *reinterpret_cast<const float4*>(ys + k * 64 + j);
            acc[k] = fmaf(yv.x, xv[0], acc[k]); acc[k] = fmaf(yv.y, xv[1], acc[k]);
            acc[k] = fmaf(yv.z, xv[2], acc[k]); acc[k] = fmaf(yv.w, xv[3], acc[k]);
          }
        }
      }
    }
  }
  if (c < C) {
#pragma unroll
    for (int k = 0; k < PW_MAXK; ++k) if (k < K) partial[((size_t)blockIdx.x * K + k) * C + c] = acc[k];
  }
}

// block = 32 outputs x 8 slice-lanes; lanes folded in a fixed order
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int nslices, int K, int C, int kc_layout, int accumulate) {
  __shared__ float a1[256];
  const int il = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il;
  float s = 0.f;
  if (i < K * C)
    for (int sidx = lane; sidx < nslices; sidx += 8) s += partial[(size_t)sidx * K * C + i];
  a1[threadIdx.x] = s;
  __syncthreads();
  if (lane != 0 || i >= K * C) return;
  for (int l = 1; l < 8; ++l) s += a1[l * 32 + il];
  const int k = i / C, c = i % C;
  const int o = kc_layout ? i : c * K + k;
  dw[o] = (accumulate ? dw[o] : 0.f) + s;
}

// rowsum over (n, p) per k : stage 1 per (n,k) row, stage 2 over n (fixed order)
__global__ __launch_bounds__(256) void hm_rowsum1_kernel(const float* __restrict__ y, float* __restrict__ part, int HW) {
  __shared__ float red[4];
  const float* row = y + (size_t)blockIdx.x * HW;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) s += row[i];
  s = block_sum<4>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void hm_rowsum2_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int K, int accumulate) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += part[n * K + k];
  out[k] = (accumulate ? out[k] : 0.f) + s;
}

static int pw_check(int N, int HW, int C, int K, int dtype, int* CH) {
  if (dtype != MI355_F32 && dtype != MI355_BF16) MI_FAIL(MI355_EINVAL, "pw: bad dtype");
  *CH = dtype == MI355_BF16 ? 8 : 4;
  if (N < 1 || HW < 1 || K < 1 || K > PW_MAXK || C % *CH) MI_FAIL(MI355_EINVAL, "pw: unsupported shape N=%d HW=%d C=%d K=%d", N, HW, C, K);
  return 0;
}

extern "C" int mi355_pw_c2k(const void* x, const float* w, const float* bias, float* y, int N, int HW, int C, int K, int w_transposed, int dtype, void* stream) {
  int CH; if (int e = pw_check(N, HW, C, K, dtype, &CH)) return e;
  const size_t esz = dtype == MI355_BF16 ? 2 : 4;
  const size_t smem = PW_PIX * (C * esz + 16) + (size_t)K * C * 4;
  if (smem > 160 * 1024) MI_FAIL(MI355_EINVAL, "pw_c2k: C=%d too large", C);
  dim3 grid(cdiv(HW, PW_PIX), N);
  if (dtype == MI355_BF16) {
    static bool a = false; if (!a) { (void)hipFuncSetAttribute((const void*)pw_c2k_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a = true; }
    hipLaunchKernelGGL(pw_c2k_kernel<bf16_t>, grid, dim3(256), smem, as_stream(stream), (const bf16_t*)x, w, bias, y, HW, C, K, w_transposed);
  } else {
    static bool a = false; if (!a) { (void)hipFuncSetAttribute((const void*)pw_c2k_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a = true; }
    hipLaunchKernelGGL(pw_c2k_kernel<float>, grid, dim3(256), smem, as_stream(stream), (const float*)x, w, bias, y, HW, C, K, w_transposed);
  }
  MI_CHECK_LAUNCH("pw_c2k");
  return MI355_OK;
}

static int pw_k2c_impl(const float* y, const float* w, const float* bias, const void* residual, const float* scale_dev, void* out,
                       int N, int HW, int C, int K, int w_transposed, int dtype, float* stat_partial, void* stream) {
  int CH; if (int e = pw_check(N, HW, C, K, dtype, &CH)) return e;
  const int tiles_per_img = cdiv(HW, PW_PIX), ntiles = N * tiles_per_img;
  static int ncu = 0;
  if (!ncu) { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); ncu = (hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
  static const int per_env = getenv("MI355_PW_PER_CU") ? atoi(getenv("MI355_PW_PER_CU")) : 0;       // experiment switch
  const size_t smem = (size_t)K * PW_PIX * 4 + (size_t)K * 32 * CH * 4 + (stat_partial ? (size_t)4 * 32 * CH * 3 * 4 : 0);
  const int cgroups = cdiv(C, 32 * CH);
  // persistent: four blocks per CU share the tiles (LDS and registers allow 4), fewer when there are fewer tiles
  int gx = cdiv(ncu * (per_env > 0 ? per_env : 4), cgroups); if (gx > ntiles) gx = ntiles; if (gx < 1) gx = 1;
  dim3 grid(gx, cgroups);
  if (dtype == MI355_BF16) hipLaunchKernelGGL(pw_k2c_kernel<bf16_t>, grid, dim3(256), smem, as_stream(stream), y, w, bias, (const bf16_t*)residual, scale_dev, (bf16_t*)out, HW, C, K, w_transposed, stat_partial, tiles_per_img, ntiles);
  else hipLaunchKernelGGL(pw_k2c_kernel<float>, grid, dim3(256), smem, as_stream(stream), y, w, bias, (const float*)residual, scale_dev, (float*)out, HW, C, K, w_transposed, stat_partial, tiles_per_img, ntiles);
  MI_CHECK_LAUNCH("pw_k2c");
  return MI355_OK;
}
extern "C" int mi355_pw_k2c(const float* y, const float* w, const float* bias, const void* residual, const float* scale_dev, void* out,
                            int N, int HW, int C, int K, int w_transposed, int dtype, void* stream) {
  return pw_k2c_impl(y, w, bias, residual, scale_dev, out, N, HW, C, K, w_transposed, dtype, nullptr, stream);
}
// the same with the BatchNorm statistics of the output from the epilogue: partial[N * ceil(HW/64)][C][n, mean, M2]
extern "C" int mi355_pw_k2c_stats(const float* y, const float* w, const float* bias, const void* residual, const float* scale_dev,
                                  void* out, int N, int HW, int C, int K, int w_transposed, int dtype, float* partial,
                                  size_t partial_bytes, int* nslices, void* stream) {
  if (!partial || !nslices) MI_FAIL(MI355_EINVAL, "pw_k2c_stats: partial / nslices must be given");
  const int ns = N * cdiv(HW, PW_PIX);
  if ((size_t)ns * C * 3 * sizeof(float) > partial_bytes) MI_FAIL(MI355_EWORKSPACE, "pw_k2c_stats: partial buffer too small");
  *nslices = ns;
  return pw_k2c_impl(y, w, bias, residual, scale_dev, out, N, HW, C, K, w_transposed, dtype, partial, stream);
}

static int pw_slices(int N, int HW, int* pps) {
  long P = (long)N * HW;
  long per = ((P + 511) / 512 + 63) / 64 * 64; if (per < 64) per = 64;
  *pps = (int)per;
  return (int)((P + per - 1) / per);
}
extern "C" size_t mi355_pw_wgrad_workspace(int N, int HW, int C, int K) {
  int pps; int ns = pw_slices(N, HW, &pps);
  return (size_t)ns * K * C * sizeof(float);
}
extern "C" int mi355_pw_wgrad(const void* x, const float* y, float* dw, int kc_layout, int accumulate, int N, int HW, int C, int K,
                              int dtype, void* ws, size_t ws_bytes, void* stream) {
  int CH; if (int e = pw_check(N, HW, C, K, dtype, &CH)) return e;
  if (HW % 64) MI_FAIL(MI355_EINVAL, "pw_wgrad: HW=%d must be a multiple of 64", HW);
  if (!ws || ws_bytes < mi355_pw_wgrad_workspace(N, HW, C, K)) MI_FAIL(MI355_EWORKSPACE, "pw_wgrad: workspace too small");
  int pps; int ns = pw_slices(N, HW, &pps);
  dim3 grid(ns, cdiv(C, 256));
  float* partial = reinterpret_cast<float*>(ws);
  if (dtype == MI355_BF16) hipLaunchKernelGGL(pw_wgrad_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)x, y, partial, N, HW, C, K, pps);
  else hipLaunchKernelGGL(pw_wgrad_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, y, partial, N, HW, C, K, pps);
  hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3(cdiv(K * C, 32)), dim3(256), 0, as_stream(stream), partial, dw, ns, K, C, kc_layout, accumulate);
  MI_CHECK_LAUNCH("pw_wgrad");
  return MI355_OK;
}

extern "C" int mi355_hm_rowsum(const float* y, float* out, int accumulate, int N, int K, int HW, void* ws, size_t ws_bytes, void* stream) {
  if (N < 1 || K < 1 || HW < 1) MI_FAIL(MI355_EINVAL, "hm_rowsum: bad shape");
  if (!ws || ws_bytes < (size_t)N * K * sizeof(float)) MI_FAIL(MI355_EWORKSPACE, "hm_rowsum: workspace too small");
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(hm_rowsum1_kernel, dim3(N * K), dim3(256), 0, as_stream(stream), y, part, HW);
  hipLaunchKernelGGL(hm_rowsum2_kernel, dim3(cdiv(K, 64)), dim3(64), 0, as_stream(stream), part, out, N, K, accumulate);
  MI_CHECK_LAUNCH("hm_rowsum");
  return MI355_OK;
}
