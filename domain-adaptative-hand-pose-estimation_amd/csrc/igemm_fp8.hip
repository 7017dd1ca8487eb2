// fp8 operand path of the implicit-GEMM convolution family (gfx950, OCP e4m3 / e5m2, fp32 accumulate, bf16 out).
//
// Why: the bf16 gather kernel is bounded by the bytes its tiles pull through the CU's vector-memory path per MFMA
// (~0.5 KB / MFMA at a sustained ~15 TB/s of tile fills).  A 128-byte LDS row carries 128 fp8 channels
// instead of 64 bf16 ones -- the same staged bytes and the same ds_read_b128 stream for twice the contraction -- and the K = 64
// instruction v_mfma_f32_32x32x64_f8f6f4 (16 passes) does that contraction at twice the rate of the K = 16 fp8 MFMA, which has
// the cycles of the bf16 one.  A lane's two 16-byte fragment reads of a row are the 32 bytes of one operand; both operands use
// the same lane -> k assignment, so any k order is consistent (profiles/mx_mfma_probe.hip).
//
// Also here: per-tensor scaled quantisation bf16 / fp32 -> fp8 with amax tracking (delayed scaling: quantise with the
// scale derived from the previous amax while recording the current one; `mi355_fp8_amax` + `mi355_fp8_update_scale` give
// the just-in-time form; every copy carries the descale it was made with, fp8_common.h), and the fp8 weight pack ([O][T][I]
// and [I][T][O], per-tensor scale).  The weight-gradient kernels on fp8 operands are in wgrad_fp8.hip.
#include "igemm_common.h"
#include "fp8_common.h"
#include <stdlib.h>

typedef int i32x8_t __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------ quantisation (helpers: fp8_common.h)
// 16 input elements per thread-iteration -> one 16-byte fp8 chunk.  amax over |x| (before scaling) into state[2] via an
// integer atomic max on the float bits (exact and order-independent).
template <typename T, bool BF8, bool WRITE>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ x, unsigned char* __restrict__ q, float* __restrict__ state, long n16,
                                                            float* __restrict__ rec) {
  constexpr int PER = Chunk<T>::N, NC = 16 / PER;
  const float scale = state[0];
  if (WRITE && rec && blockIdx.x == 0 && threadIdx.x == 0) { rec[0] = scale; rec[1] = state[1]; rec[2] = 0.f; rec[3] = 0.f; }   // this copy's own record
  const unsigned seen = fp8_amax_seen(state);
  float amax = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    float v[16];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float w[PER]; Chunk<T>::load(x + i * 16 + c * PER, w);
#pragma unroll
      for (int e = 0; e < PER; ++e) v[c * PER + e] = w[e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) amax = fmaxf(amax, fabsf(v[e]));      // (fmaxf ignores NaN)
    if constexpr (WRITE) {
      uint4 o;
      o.x = pack4_fp8<BF8>(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
      o.y = pack4_fp8<BF8>(v[4] * scale, v[5] * scale, v[6] * scale, v[7] * scale);
      o.z = pack4_fp8<BF8>(v[8] * scale, v[9] * scale, v[10] * scale, v[11] * scale);
      o.w = pack4_fp8<BF8>(v[12] * scale, v[13] * scale, v[14] * scale, v[15] * scale);
      reinterpret_cast<uint4*>(q)[i] = o;
    }
  }
  fp8_record_amax(amax, state, seen);
}

// scale := fmt_max / (amax * 2^margin) (1 when nothing was seen yet), descale := 1 / scale, amax := 0.  `n` states.
__global__ void fp8_update_scale_kernel(float* __restrict__ states, int n, int stride, float fmt_max, float margin_pow2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* s = states + (size_t)i * stride;
  const float amax = __uint_as_float(reinterpret_cast<unsigned*>(s)[2]);
  if (amax > 0.f && amax < 3.0e38f) {
    // power-of-two scale: scaling / descaling are exact, only the fp8 rounding itself loses bits
    const float raw = fmt_max / (amax * margin_pow2);
    const float sc = exp2f(floorf(log2f(raw)));
    s[0] = sc; s[1] = 1.0f / sc;
  } else if (s[0] == 0.f) { s[0] = 1.f; s[1] = 1.f; }
  reinterpret_cast<unsigned*>(s)[2] = 0u;
}

extern "C" int mi355_fp8_quantize(const void* x, void* q, float* state, long n, int src_dtype, int fmt, int write, void* stream) {
  if (!x || !state || (write && !q) || n < 16 || n % 16) MI_FAIL(MI355_EINVAL, "fp8_quantize: n=%ld must be a positive multiple of 16", n);
  if (fmt != 0 && fmt != 1) MI_FAIL(MI355_EINVAL, "fp8_quantize: fmt %d (0 = e4m3, 1 = e5m2)", fmt);
  if (write < 0 || write > 2) MI_FAIL(MI355_EINVAL, "fp8_quantize: write %d (0 = amax only, 1 = copy, 2 = copy + its 16-byte scale record)", write);
  if (src_dtype != MI355_BF16 && src_dtype != MI355_F32) MI_FAIL(MI355_EINVAL, "fp8_quantize: source dtype %d", src_dtype);
  const long n16 = n / 16;
  int grid = (int)((n16 + 255) / 256); if (grid > 1024) grid = 1024;
  hipStream_t st = as_stream(stream);
  unsigned char* o = reinterpret_cast<unsigned char*>(q);
  float* rec = write == 2 ? reinterpret_cast<float*>(o + n) : nullptr;      // write = 2: q has 16 more bytes, the copy's {scale, descale, 0, 0}
#define MI_Q(T, BF8, W) hipLaunchKernelGGL((quantize_fp8_kernel<T, BF8, W>), dim3(grid), dim3(256), 0, st, (const T*)x, o, state, n16, rec)
  if (src_dtype == MI355_BF16) {
    if (!write) MI_Q(bf16_t, false, false); else if (fmt) MI_Q(bf16_t, true, true); else MI_Q(bf16_t, false, true);
  } else {
    if (!write) MI_Q(float, false, false); else if (fmt) MI_Q(float, true, true); else MI_Q(float, false, true);
  }
#undef MI_Q
  MI_CHECK_LAUNCH("fp8_quantize");
  return MI355_OK;
}

extern "C" int mi355_fp8_update_scale(float* states, int n, int stride_floats, int fmt, int margin, void* stream) {
  if (!states || n < 1 || stride_floats < 3 || (fmt != 0 && fmt != 1) || margin < 0 || margin > 8)
    MI_FAIL(MI355_EINVAL, "fp8_update_scale: bad args");
  hipLaunchKernelGGL(fp8_update_scale_kernel, dim3(cdiv(n, 64)), dim3(64), 0, as_stream(stream), states, n, stride_floats,
                     fmt ? 57344.f : 448.f, (float)(1 << margin));
  MI_CHECK_LAUNCH("fp8_update_scale");
  return MI355_OK;
}

// fp32 master [O][T][I] (conv-form) -> e4m3 wf [O][T][I] and wt [I][T][O], both * state[0].  I, O multiples of 32.
// Also records amax(|w|) in state[2] (block-level atomic) for the next scale update (delayed scaling).
__device__ __forceinline__ void pack_fp8_block(const float* __restrict__ w, unsigned char* __restrict__ wf, unsigned char* __restrict__ wt,
                                               float* __restrict__ state, int O, int T, int I, int b) {
  __shared__ float tile[32][33];
  __shared__ float red[4];
  const float scale = state[0];
  const unsigned seen = fp8_amax_seen(state);
  if (b == 0 && threadIdx.x == 0) state[3] = state[1];      // the descale that belongs to THIS packed copy (fp8_common.h)
  const int tiles_i = I / 32, tiles_o = O / 32;
  const int tap = b / (tiles_i * tiles_o), r = b % (tiles_i * tiles_o);
  const int o0 = (r / tiles_i) * 32, i0 = (r % tiles_i) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int o = o0 + ty + 8 * k;
    const float v = w[((size_t)o * T + tap) * I + i0 + tx];
    amax = fmaxf(amax, fabsf(v));
    tile[ty + 8 * k][tx] = v * scale;
  }
  amax = block_max<4>(amax, red);                     // (contains the barriers that publish `tile`)
  if (threadIdx.x == 0 && amax > 0.f && __float_as_uint(amax) > seen) (void)atomicMax(reinterpret_cast<unsigned*>(state) + 2, __float_as_uint(amax));
  __syncthreads();
  {   // wf rows: each thread packs 4 consecutive i of one o
    const int o = threadIdx.x >> 3, g = threadIdx.x & 7;
    const unsigned v = pack4_fp8<false>(tile[o][4 * g], tile[o][4 * g + 1], tile[o][4 * g + 2], tile[o][4 * g + 3]);
    *reinterpret_cast<unsigned*>(wf + ((size_t)(o0 + o) * T + tap) * I + i0 + 4 * g) = v;
  }
  {   // wt rows: 4 consecutive o of one i
    const int i = threadIdx.x >> 3, g = threadIdx.x & 7;
    const unsigned v = pack4_fp8<false>(tile[4 * g][i], tile[4 * g + 1][i], tile[4 * g + 2][i], tile[4 * g + 3][i]);
    *reinterpret_cast<unsigned*>(wt + ((size_t)(i0 + i) * T + tap) * O + o0 + 4 * g) = v;
  }
}
__global__ __launch_bounds__(256) void pack_weights_fp8_kernel(const float* __restrict__ w, unsigned char* __restrict__ wf,
                                                                unsigned char* __restrict__ wt, float* __restrict__ state,
                                                                int O, int T, int I) {
  pack_fp8_block(w, wf, wt, state, O, T, I, blockIdx.x);
}
// every fp8 conv weight of an optimizer group in ONE launch (delayed scales): items in device memory, block -> item by
// binary search over the first-block prefix (as mi355_pack_weights_batched)
__global__ __launch_bounds__(256) void pack_weights_fp8_batched_kernel(const mi355_pack8_item* __restrict__ items, int nitems) {
  int lo = 0, hi = nitems - 1;
  const int b = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].blk0 <= b) lo = mid; else hi = mid - 1; }
  const mi355_pack8_item it = items[lo];
  pack_fp8_block(it.w, (unsigned char*)it.wf, (unsigned char*)it.wt, it.state, it.O, it.T, it.I, b - it.blk0);
}
extern "C" int mi355_pack_weights_fp8_batched(const mi355_pack8_item* items_dev, int nitems, int total_blocks, void* stream) {
  if (!items_dev || nitems < 1 || total_blocks < 1) MI_FAIL(MI355_EINVAL, "pack_weights_fp8_batched: bad args");
  hipLaunchKernelGGL(pack_weights_fp8_batched_kernel, dim3(total_blocks), dim3(256), 0, as_stream(stream), items_dev, nitems);
  MI_CHECK_LAUNCH("pack_weights_fp8_batched");
  return MI355_OK;
}

extern "C" int mi355_pack_weights_fp8(const float* w_master, void* wf, void* wt, float* state, int O, int T, int I, int margin, void* stream) {
  if (!w_master || !wf || !wt || !state || O < 32 || I < 32 || O % 32 || I % 32 || T < 1)
    MI_FAIL(MI355_EINVAL, "pack_weights_fp8: O=%d I=%d must be multiples of 32 (T=%d)", O, I, T);
  if (margin >= 0) {      // just-in-time scale from the amax of this very tensor (two more launches)
    if (int e = mi355_fp8_quantize(w_master, nullptr, state, (long)O * T * I, MI355_F32, 0, 0, stream)) return e;   // amax only
    if (int e = mi355_fp8_update_scale(state, 1, 4, 0, margin, stream)) return e;
  }                       // margin < 0: the scale already in `state` (delayed scaling); the amax of w is recorded either way
  hipLaunchKernelGGL(pack_weights_fp8_kernel, dim3((O / 32) * (I / 32) * T), dim3(256), 0, as_stream(stream), w_master,
                     (unsigned char*)wf, (unsigned char*)wt, state, O, T, I);
  MI_CHECK_LAUNCH("pack_weights_fp8");
  return MI355_OK;
}

// ------------------------------------------------------------------------------------ gather GEMM, fp8 operands
template <int BM, int BN, bool KW3 = false>
struct Fp8Smem {
  static constexpr int kARows = KW3 ? BM + 4 * (BM / 8) + 4 : BM;        // KW3: segmented A image with spare rows (as GatherSmem)
  static constexpr int kStage = (kARows + BN) * 128;
  static constexpr int kOutStride = BN * 2 + 16;             // bf16 output tile rows
  static constexpr int kOut = BM * kOutStride;
  static constexpr int kBytes = (kStage > kOut ? kStage : kOut) + BM * 4;
};

// 4 waves (2 x 2), each a (BM/2) x (BN/2) sub-tile of 32x32 MFMA blocks.  K-tile = 128 channels of one tap
// (8 chunks of 16 bytes); register-staged pipeline over one LDS stage, like the bf16 kernel's default path.
// A_BF8: the gathered operand is e5m2 (gradients), the weights are always e4m3.  EPI 1: BatchNorm statistics of the output.
// KW3: 3x3 / unit-stride layers (forward and input gradient): the three taps of a kernel row read the same pixels shifted by
// -1 / 0 / +1, so one staged A tile per (kernel row, 128-channel chunk) serves three K sub-steps -- the construction of the bf16
// kernel's KW3 path (igemm.hip), whose 128-byte rows carry 64 bf16 channels where these carry 128 fp8 channels: same LDS image,
// same shifted fragment reads.  With the K=64 MFMA these layers are bounded by the tile fills; this takes a third of them away.
template <int BM, int BN, bool A_BF8, int EPI, bool KW3 = false>
__global__ __launch_bounds__(256) void gather_fp8_kernel(const GatherArgs p) {
  constexpr int CHI = 16;                                     // fp8 elements per 16-byte chunk
  constexpr int CHO = 8;                                      // bf16 output elements per chunk
  constexpr int NTHR = 256, RPP = 32;
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int RA = BM / RPP, RB = BN / RPP;
  using SM = Fp8Smem<BM, BN, KW3>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* row_off = reinterpret_cast<int*>(smem + SM::kBytes - BM * 4);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int tile_g = xcd_remap(blockIdx.x, p.ntiles);
  const int phi = tile_g % p.nphase, tile = tile_g / p.nphase;
  const Phase& P = p.ph[phi];
  if (tile >= P.ntm * p.ntn) return;
  const int pM = P.M, pOHp = P.OHp, pOWp = P.OWp, pkchunks = P.ntaps << p.cshift;
  const Tap* __restrict__ ptaps = p.taps + P.tap0;
  const int m0 = (tile / p.ntn) * BM, n0 = (tile % p.ntn) * BN;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lc = t & 7, lr = t >> 3;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);

  int iy0[RA], ix0[RA], pix0[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + lr + RPP * i;
    if (m < pM) {
      const int ox = m % pOWp, r = m / pOWp, oy = r % pOHp, n = r / pOHp;
      iy0[i] = oy * p.in_sy; ix0[i] = ox * p.in_sx;
      pix0[i] = (n * p.Hi * p.Wi + iy0[i] * p.Wi + ix0[i]) * p.Ci;
    } else { iy0[i] = -(1 << 20); ix0[i] = 0; pix0[i] = 0; }
  }
  if (t < BM) {
    const int m = m0 + t; int off = -1;
    if (m < pM) {
      const int ox = m % pOWp, r = m / pOWp, oy = r % pOHp, n = r / pOHp;
      off = ((n * p.Ho + oy * p.out_sy + P.out_oy) * p.Wo + ox * p.out_sx + P.out_ox) * p.ldd;
    }
    row_off[t] = off;
  }
  const int cmask = (1 << p.cshift) - 1;
  uint4 ra0[RA], rb0[RB];
  auto load_tile = [&](int kt, uint4 (&ra)[RA], uint4 (&rb)[RB]) {
    const Tap tp = ptaps[(kt * 8) >> p.cshift];
    const int cc = (((kt * 8) & cmask) + lc) * CHI;
    const int toff = ((int)tp.dy * p.Wi + (int)tp.dx) * p.Ci + cc;
    const int koff = (int)tp.widx * p.Ci + cc;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int iy = iy0[i] + tp.dy, ix = ix0[i] + tp.dx;
      const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      ra[i] = buf_load16(rsA, ok ? pix0[i] + toff : OOB_OFF);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int n = n0 + lr + RPP * i;
      rb[i] = buf_load16(rsB, n < p.Nout ? n * p.ldb + koff : OOB_OFF);
    }
  };
  char* as = smem;
  char* bs = smem + SM::kARows * 128;
  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int r31 = lane & 31, hi = lane >> 5;

  if constexpr (KW3) {
    // sub-step ks = (group g, kw), group g = (kernel row kh, 128-channel chunk): tap index kh*3 + kw
    const int nchunk = p.Ci >> 7, nsub = 9 * nchunk;
    for (int i = t; i < SM::kARows * 8; i += NTHR) reinterpret_cast<uint4*>(as)[i] = make_uint4(0, 0, 0, 0);
    int simg[RA];                                         // LDS byte offset of this thread's A rows in the segmented image
#pragma unroll
    for (int i = 0; i < RA; ++i) { const int r = lr + RPP * i; simg[i] = swz128(r + 2 + 4 * (r >> p.lw), lc); }
    int rimg[MT];                                         // image row of this lane's fragment rows
#pragma unroll
    for (int i = 0; i < MT; ++i) { const int q = wm0 + i * 32 + r31; rimg[i] = q + 2 + 4 * (q >> p.lw); }
    auto load_a = [&](int g) {
      const int kh = g / nchunk, chunk = g - kh * nchunk;
      const int dyv = ptaps[kh * 3].dy;
      const int toff = dyv * p.Wi * p.Ci + chunk * 128 + lc * CHI;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const bool ok = (unsigned)(iy0[i] + dyv) < (unsigned)p.Hi;
        ra0[i] = buf_load16(rsA, ok ? pix0[i] + toff : OOB_OFF);
      }
    };
    auto load_b = [&](int ks) {
      const int g = ks / 3, kw = ks - g * 3;
      const int kh = g / nchunk, chunk = g - kh * nchunk;
      const int koff = (int)ptaps[kh * 3 + kw].widx * p.Ci + chunk * 128 + lc * CHI;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int n = n0 + lr + RPP * i;
        rb0[i] = buf_load16(rsB, n < p.Nout ? n * p.ldb + koff : OOB_OFF);
      }
    };
    load_a(0);
    load_b(0);
    int kw = 0, g = 0;
    for (int ks = 0; ks < nsub; ++ks) {
      const int dxv = ptaps[(g / nchunk) * 3 + kw].dx;
      __syncthreads();                                    // sub-step ks-1 fully multiplied
      if (kw == 0) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<uint4*>(as + simg[i]) = ra0[i];
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) *reinterpret_cast<uint4*>(bs + swz128(lr + RPP * i, lc)) = rb0[i];
      __syncthreads();
      if (ks + 1 < nsub) {
        load_b(ks + 1);
        if (kw == 2) load_a(g + 1);                       // the next group's A tile travels during this group's last sub-step
      }
      __builtin_amdgcn_s_setprio(1);
      int arow[MT], axor[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) { const int r = rimg[i] + dxv; arow[i] = r * 128; axor[i] = (r >> 1) & 7; }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint4 a[MT][2], b[NT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int c = 0; c < 2; ++c) a[i][c] = *reinterpret_cast<const uint4*>(as + arow[i] + (((4 * u + 2 * hi + c) ^ axor[i]) << 4));
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int c = 0; c < 2; ++c) b[j][c] = *reinterpret_cast<const uint4*>(bs + swz128(wn0 + j * 32 + r31, 4 * u + 2 * hi + c));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const i32x8_t av = {(int)a[i][0].x, (int)a[i][0].y, (int)a[i][0].z, (int)a[i][0].w, (int)a[i][1].x, (int)a[i][1].y, (int)a[i][1].z, (int)a[i][1].w};
            const i32x8_t bv = {(int)b[j][0].x, (int)b[j][0].y, (int)b[j][0].z, (int)b[j][0].w, (int)b[j][1].x, (int)b[j][1].y, (int)b[j][1].z, (int)b[j][1].w};
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, av, acc[i][j], 0, A_BF8 ? 1 : 0, 0, 0, 0, 0);
          }
      }
      __builtin_amdgcn_s_setprio(0);
      if (++kw == 3) { kw = 0; ++g; }
    }
  } else {
  const int nk = (pkchunks + 7) >> 3;
  load_tile(0, ra0, rb0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<uint4*>(as + swz128(lr + RPP * i, lc)) = ra0[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<uint4*>(bs + swz128(lr + RPP * i, lc)) = rb0[i];
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1, ra0, rb0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // two 16-byte reads per fragment row = the 32 bytes a lane gives v_mfma_f32_32x32x64_f8f6f4 (k set: chunks 4u + 2hi and
      // 4u + 2hi + 1 of the 128-byte row).  The K=64 instruction does the work of four 32x32x16 fp8 MFMAs in the cycles of two
      // (16 passes against 4 x 8; profiles/mx_mfma_probe.hip: same bits, 3.8 against 1.9 PFLOP/s).  It is the block-scaled
      // instruction with both scale operands the literal 0, which the compiler emits as the unscaled opcode.
      uint4 a[MT][2], b[NT][2];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) a[i][c] = *reinterpret_cast<const uint4*>(as + swz128(wm0 + i * 32 + r31, 4 * u + 2 * hi + c));
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) b[j][c] = *reinterpret_cast<const uint4*>(bs + swz128(wn0 + j * 32 + r31, 4 * u + 2 * hi + c));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const i32x8_t av = {(int)a[i][0].x, (int)a[i][0].y, (int)a[i][0].z, (int)a[i][0].w, (int)a[i][1].x, (int)a[i][1].y, (int)a[i][1].z, (int)a[i][1].w};
          const i32x8_t bv = {(int)b[j][0].x, (int)b[j][0].y, (int)b[j][0].z, (int)b[j][0].w, (int)b[j][1].x, (int)b[j][1].y, (int)b[j][1].z, (int)b[j][1].w};
          // operands swapped (weights first): acc holds D^T, lane = output pixel, registers = runs of 4 channels.
          // cbsz / blgp = formats of the first / second operand: 0 = e4m3, 1 = e5m2
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, av, acc[i][j], 0, A_BF8 ? 1 : 0, 0, 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
  }
  }
  __syncthreads();

  // ---- epilogue: (acc * descale [* lambda] + bias) -> bf16 -> LDS tile -> coalesced 16-byte rows (+residual / +dx)
  char* outs = smem;
  const float scale = (p.scale ? *p.scale : 1.0f) * (p.scale2 ? *p.scale2 : 1.0f) * (p.scale3 ? *p.scale3 : 1.0f);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ml = wm0 + i * 32 + r31;
        const int nl = wn0 + j * 32 + 8 * g + 4 * hi;
        union { bf16_t h[4]; uint2 q; } u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float b = (p.bias && (n0 + nl + e) < p.Nout) ? p.bias[n0 + nl + e] : 0.f;
          u.h[e] = (bf16_t)(acc[i][j][4 * g + e] * scale + b);
        }
        *reinterpret_cast<uint2*>(outs + ml * SM::kOutStride + nl * 2) = u.q;
      }
  __syncthreads();
  constexpr int CPR = BN / CHO;
  bf16_t* __restrict__ D = reinterpret_cast<bf16_t*>(p.D);
  const bf16_t* __restrict__ R = reinterpret_cast<const bf16_t*>(p.residual);
  float sn = 0.f, smean[CHO], sm2[CHO];
#pragma unroll
  for (int e = 0; e < CHO; ++e) { smean[e] = 0.f; sm2[e] = 0.f; }
  for (int id = t; id < BM * CPR; id += NTHR) {
    const int r = id / CPR, c = id % CPR;
    const int off = row_off[r];
    const int n = n0 + c * CHO;
    if (off < 0 || n >= p.Nout) continue;
    float v[CHO];
    Chunk<bf16_t>::load(reinterpret_cast<const bf16_t*>(outs + r * SM::kOutStride + c * 16), v);
    const size_t g = (size_t)off + n;
    if constexpr (EPI == 1) {
      sn += 1.f; const float inv = 1.f / sn;
#pragma unroll
      for (int e = 0; e < CHO; ++e) { const float d = v[e] - smean[e]; smean[e] += d * inv; sm2[e] += d * (v[e] - smean[e]); }
    }
    if (R) { float w[CHO]; Chunk<bf16_t>::load(R + g, w);
#pragma unroll
      for (int e = 0; e < CHO; ++e) v[e] += w[e]; }
    if (p.accumulate) { float w[CHO]; Chunk<bf16_t>::load(D + g, w);
#pragma unroll
      for (int e = 0; e < CHO; ++e) v[e] += w[e]; }
    Chunk<bf16_t>::store(D + g, v);
  }
  if constexpr (EPI == 1) {
    // same fold as the bf16 kernel: row lanes of a wave by shuffle-down (lower lane = left operand), the four waves through
    // LDS in wave order -> (n, mean, M2) per channel and m-tile slice, fixed order
    constexpr int NW = NTHR / 64;
    static_assert(CPR <= 32, "chunk columns of one tile row must fit half a wave");
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1) {
      const float nb = __shfl_down(sn, o, 64);
      const float nt = sn + nb, f = nt > 0.f ? nb / nt : 0.f;
#pragma unroll
      for (int e = 0; e < CHO; ++e) {
        const float mb = __shfl_down(smean[e], o, 64), vb = __shfl_down(sm2[e], o, 64);
        const float d = mb - smean[e];
        smean[e] += d * f; sm2[e] += vb + d * d * sn * f;
      }
      sn = nt;
    }
    __syncthreads();
    float* sp = reinterpret_cast<float*>(smem);        // [NW][BN][3]
    static_assert(NW * BN * 3 * 4 <= SM::kBytes - BM * 4, "statistics scratch must fit in the tile staging area");
    if (lane < CPR) {
      const int c = t % CPR;
#pragma unroll
      for (int e = 0; e < CHO; ++e) {
        float* q = sp + ((size_t)wave * BN + c * CHO + e) * 3;
        q[0] = sn; q[1] = smean[e]; q[2] = sm2[e];
      }
    }
    __syncthreads();
    if (t < BN && n0 + t < p.Nout) {
      float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float* q = sp + ((size_t)w * BN + t) * 3;
        const float nb = q[0];
        if (nb > 0.f) { const float nt = n + nb, f = nb / nt, d = q[1] - mean; mean += d * f; m2 += q[2] + d * d * n * f; n = nt; }
      }
      const int slice = (tile / p.ntn) * p.nphase + phi;
      float* out = p.stat_partial + ((size_t)slice * p.Nout + n0 + t) * 3;
      out[0] = n; out[1] = mean; out[2] = m2;
    }
  }
}

template <int BM, int BN, bool KW3 = false>
static void launch_fp8(GatherArgs& a, hipStream_t st) {
  constexpr int smem = Fp8Smem<BM, BN, KW3>::kBytes;
  a.ntn = cdiv(a.Nout, BN);
  int mx = 0;
  for (int i = 0; i < a.nphase; ++i) { a.ph[i].ntm = cdiv(a.ph[i].M, BM); if (a.ph[i].ntm > mx) mx = a.ph[i].ntm; }
  a.ntiles = a.nphase * mx * a.ntn;
  a.stat_slices = 0;
  if (a.stat_partial) {
    bool even = !a.residual && !a.accumulate;
    for (int i = 0; i < a.nphase; ++i) even = even && a.ph[i].ntm == mx;
    if (even && (size_t)a.nphase * mx * a.Nout * 3 * sizeof(float) <= a.stat_bytes) a.stat_slices = a.nphase * mx;
    else a.stat_partial = nullptr;
  }
#define MI_L(BF8, EPI) do { auto kern = gather_fp8_kernel<BM, BN, BF8, EPI, KW3>; static bool set_ = false; \
    if (!set_) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); set_ = true; } \
    hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(256), smem, st, a); } while (0)
  if (a.a_fmt) { if (a.stat_partial) MI_L(true, 1); else MI_L(true, 0); }
  else { if (a.stat_partial) MI_L(false, 1); else MI_L(false, 0); }
#undef MI_L
}

static long g_fp8_kw3_min = -1;      // run-time switch (mi355_set_fp8_kw3); -1: the environment decides (MI355_FP8_KW3, default 1024)
extern "C" long mi355_set_fp8_kw3(long min_tiles) { const long prev = g_fp8_kw3_min; g_fp8_kw3_min = min_tiles < 0 ? -1 : min_tiles; return prev; }

static int ilog2x(int v) { int s = 0; while ((1 << s) < v) ++s; return ((1 << s) == v) ? s : -1; }

int dispatch_gather_fp8(GatherArgs& a, hipStream_t st) {
  if (a.Ci % 128) MI_FAIL(MI355_EINVAL, "fp8 gather: the contraction channels (%d) must be a multiple of 128", a.Ci);
  a.cshift = ilog2x(a.Ci / 16);
  if (a.cshift < 3) MI_FAIL(MI355_EINVAL, "fp8 gather: Ci/16 must be a power of two >= 8 (Ci=%d)", a.Ci);
  if (a.Nout % 8) MI_FAIL(MI355_EINVAL, "fp8 gather: Nout=%d not a multiple of 8", a.Nout);
  if (a.nphase < 1 || a.nphase > 4) MI_FAIL(MI355_EINVAL, "fp8 gather: nphase=%d", a.nphase);
  if (a.bnb_partial) MI_FAIL(MI355_EINVAL, "fp8 gather: no BatchNorm-backward epilogue in this build");
  long Mtot = 0, ntaps_tot = 0; double flops = 0.0;
  for (int i = 0; i < a.nphase; ++i) {
    Mtot += a.ph[i].M; ntaps_tot += a.ph[i].ntaps;
    flops += 2.0 * a.ph[i].M * (double)a.Nout * a.ph[i].ntaps * a.Ci;
  }
  const long imgs = a.ph[0].M / ((long)a.ph[0].OHp * a.ph[0].OWp);
  const long abytes = imgs * a.Hi * a.Wi * a.Ci, bbytes = (long)a.Nout * a.ldb;
  if (abytes >= (1L << 31) || bbytes >= (1L << 31) || Mtot * a.Nout * 2 >= (1L << 31))
    MI_FAIL(MI355_EINVAL, "fp8 gather: tensor too large for 32-bit byte offsets");
  a.a_bytes = (unsigned)abytes; a.b_bytes = (unsigned)bbytes;
  ProfScope ps(st, flops, (double)abytes + (double)bbytes * ntaps_tot / (a.ldb / a.Ci) + (double)Mtot * a.Nout * 2);
  static const int force = getenv("MI355_FP8_TILE") ? atoi(getenv("MI355_FP8_TILE")) : -1;
  const long t128 = cdiv(Mtot, 128L) * cdiv(a.Nout, 128);
  // 3x3 / unit stride / same-size maps of a power-of-two width <= 128: the A-tile-sharing variant (KW3 above; conditions as in
  // dispatch_gather of igemm.hip) from 1024 128x128 tiles on (MI355_FP8_KW3 / mi355_set_fp8_kw3: 0 off, n = smallest tile count).
  // B=64: 3x3 256->256 @64x64 183 -> 168 us, @32x32 53.5 -> 48.4; below 1024 tiles neutral to slower (@16x16 18.9 -> 21.0).
  // Iteration: ResNet-101 512x512 76.45 / 76.60 -> 76.24 / 76.05 ms, ResNet-50 32.11 / 32.06 -> 31.92 / 32.04.
  static const long kw3_env = getenv("MI355_FP8_KW3") ? atol(getenv("MI355_FP8_KW3")) : 1024;
  const long kw3_min = g_fp8_kw3_min >= 0 ? g_fp8_kw3_min : kw3_env;
  bool kw3 = kw3_min > 0 && t128 >= kw3_min && a.nphase == 1 && a.ph[0].ntaps == 9 && a.in_sx == 1 && a.in_sy == 1 && a.out_sx == 1 &&
             a.out_sy == 1 && a.ph[0].OWp == a.Wi && a.ph[0].OHp == a.Hi && a.Wo == a.Wi && a.Ho == a.Hi && a.Wi >= 8 && a.Wi <= 128 &&
             ilog2x(a.Wi) >= 0 && a.Nout > 64;
  for (int g = 0; g < 3 && kw3; ++g) {
    const Tap* tp = a.taps + a.ph[0].tap0 + 3 * g;
    int seen = 0;
    for (int k = 0; k < 3; ++k) { if (tp[k].dy != tp[0].dy || tp[k].dx < -1 || tp[k].dx > 1) kw3 = false; else seen |= 1 << (tp[k].dx + 1); }
    if (seen != 7 || tp[0].dy < -1 || tp[0].dy > 1) kw3 = false;
  }
  if (kw3) { a.lw = ilog2x(a.Wi); launch_fp8<128, 128, true>(a, st); }
  else if (force == 0 || (force < 0 && t128 >= 512 && a.Nout > 64)) launch_fp8<128, 128>(a, st);
  else if (force == 1 || (force < 0 && a.Nout > 64 && cdiv(Mtot, 64L) * cdiv(a.Nout, 128) >= 256)) launch_fp8<64, 128>(a, st);
  else launch_fp8<64, 64>(a, st);
  MI_CHECK_LAUNCH("gather_fp8");
  return MI355_OK;
}
