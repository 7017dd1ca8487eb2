// Implicit-GEMM convolution family on MFMA (gfx950):
//   gather-GEMM  : D[m][n] = sum_{tap} sum_c A[pix(m)+off(tap)][c] * B[n][tap][c]
//                  -> conv-form forward, and (per stride phase) conv-form dgrad / ConvTranspose fwd
//   wgrad-GEMM   : dW[o][tap][c] = sum_m DY[m][o] * X[pix(m)+off(tap)][c]   (split over m, slab reduce)
// Layout: activations NHWC, K (channels) contiguous, 16-byte chunks; tiles staged through LDS in
// 128-byte rows with an XOR swizzle so that ds_read_b128 fragment reads are conflict-free.
// bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate.  f32: v_mfma_f32_32x32x2_f32 (exact fp32, parity path).
#include "common.h"
#include <stdlib.h>

#include "igemm_common.h"

template <typename T> struct MmaTraits;
template <> struct MmaTraits<bf16_t> { static constexpr int BK = 64; static constexpr int CH = 8; };
template <> struct MmaTraits<float> { static constexpr int BK = 32; static constexpr int CH = 4; };

#define GATHER_STAGES 1     // one LDS stage: the pipeline depth lives in registers (see the K loop)
template <typename T, int BM, int BN, int STAGES = 1, bool KW3 = false>
struct GatherSmem {
  static constexpr int kARows = KW3 ? BM + 4 * (BM / 8) + 4 : BM;        // KW3: segmented A image with spare rows (see the kernel)
  static constexpr int kStage = (kARows + BN) * 128;
  static constexpr int kOutStride = BN * (int)sizeof(T) + 16;
  static constexpr int kOut = BM * kOutStride;
  static constexpr int kBytes = (STAGES * kStage > kOut ? STAGES * kStage : kOut) + BM * 4;
};

// WGM x WGN waves per workgroup (64 lanes each); every wave owns a (BM/WGM) x (BN/WGN) sub-tile.
// HM_OUT: the result is written as NCHW fp32 heat-maps [image][Nout][hw] (the 1x1 conv to the K=21 key-point maps).
// DMA: K-tiles travel global -> LDS directly (buffer_load ... lds, 1 KiB per wave-instruction, out-of-range lanes deliver
//      zeros) into a 2-stage ring: no staging registers, no ds_write, one barrier per K-tile.
// EPI: epilogue extras, compiled separately so the plain kernel keeps its register budget (3 blocks per CU):
//      0 plain, 1 + BatchNorm statistics of the output, 2 + BatchNorm-backward reduction of the output.
// KW3: 3x3 / unit-stride convs (forward and input gradient).  The three taps of a kernel row read the same pixels shifted by
//      -1 / 0 / +1, so ONE staged A tile per (kernel row, 64-channel chunk) serves three K sub-steps (three B tiles): a third
//      fewer bytes through L1 and LDS per MFMA.  The LDS A image keeps every run of min(W,128) pixels in a segment of its
//      own with spare zero rows between segments, so a shifted fragment read sees zeros exactly at the image border
//      (same construction as wgrad_kw_kernel).
// CAT: one more K tile behind the conv's own taps, gathered from a second operand pair (A2 [M][c2] at the output resolution,
//      B2 [Nout][c2]; channels c2 .. BK-1 of that tile are out-of-range lanes = zeros): y = conv(x, w) + A2 * B2^T as ONE GEMM.
// KG: K groups inside the workgroup (intra-workgroup split-K; LDS-DMA builds).  A launch that offers one tile per CU or fewer runs
//      one wave per SIMD, and its K loop is a chain of exposed load latencies (one tile in flight per CU).  With KG = 2 the block has
//      2 x WGM x WGN waves: group g multiplies its half of the K tiles out of a ring of its own (two tiles in flight per CU, two
//      waves per SIMD), group 1 then hands its accumulators to group 0 through LDS (fp32, its own ring's space) and group 0 runs
//      the usual epilogue -- streamed out by all the block's threads.  No fp32 partials in HBM.
template <typename T, int BM, int BN, bool SMALL_C, int WGM = 2, int WGN = 2, bool HM_OUT = false, bool DMA = false, int EPI = 0,
          bool KW3 = false, bool CAT = false, int KG = 1>
__global__ __launch_bounds__(64 * WGM * WGN * KG, (KW3 && BM == 256) ? 2 : 1) void gather_gemm_kernel(const GatherArgs p) {
  constexpr int CH = MmaTraits<T>::CH;
  constexpr int GTHR = 64 * WGM * WGN;                          // threads of one K group (= the whole block unless KG > 1)
  constexpr int NTHR = GTHR * KG, RPP = GTHR / 8;               // rows staged per pass (8 lanes = one 128-byte row)
  constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NT = WN / 32;
  constexpr int RA = BM / RPP, RB = BN / RPP;
  using SM = GatherSmem<T, BM, BN, DMA ? 2 : 1, KW3>;
  static_assert(!KW3 || (!DMA && !SMALL_C && !HM_OUT && sizeof(T) == 2), "KW3 is a bf16 variant of the plain large-channel kernel");
  static_assert(!DMA || !SMALL_C, "the LDS-DMA pipeline is built for the large-channel path");
  static_assert(!CAT || (!SMALL_C && !HM_OUT && !KW3 && EPI != 2), "CAT: plain large-channel forward (register-staged or LDS-DMA)");
  static_assert(KG == 1 || (KG == 2 && DMA && !CAT && !HM_OUT && EPI != 2 && BM * BN * 4 <= 2 * SM::kStage), "KG: two K groups on the LDS-DMA ring");
  constexpr int kRing = 2 * SM::kStage;                         // (KG > 1) LDS of one K group's ring; group g's ring starts at g * kRing
  constexpr int kTotal = SM::kBytes + (KG - 1) * kRing;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* row_off = reinterpret_cast<int*>(smem + kTotal - BM * 4);

  const int t = threadIdx.x, lane = t & 63;
  const int tg = KG > 1 ? t % GTHR : t, grp = KG > 1 ? __builtin_amdgcn_readfirstlane(t / GTHR) : 0;      // thread in its K group, K group
  const int wave = tg >> 6, wave_blk = t >> 6;                  // wave in its K group (sub-tile, staging), wave in the block (statistics fold)
  const int tile_g = xcd_remap(blockIdx.x, p.ntiles);
  const int phi = tile_g % p.nphase, tile = tile_g / p.nphase;
  const Phase& P = p.ph[phi];
  if (tile >= P.ntm * p.ntn) return;                 // phases of unequal size (odd extents): block-uniform exit
  const int pM = P.M, pOHp = P.OHp, pOWp = P.OWp, pkchunks = P.ntaps << p.cshift;
  const Tap* __restrict__ ptaps = p.taps + P.tap0;
  const int m0 = (tile / p.ntn) * BM, n0 = (tile % p.ntn) * BN;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int lc = tg & 7, lr = tg >> 3;

  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  const __amdgpu_buffer_rsrc_t rsA2 = make_rsrc(CAT ? p.A2 : p.A, CAT ? p.a2_bytes : 0u), rsB2 = make_rsrc(CAT ? p.B2 : p.B, CAT ? p.b2_bytes : 0u);
  const int nk_main = (pkchunks + 7) >> 3;            // K tiles of the conv's own taps; the CAT tile is K tile number nk_main

  // decode this thread's A rows once
  int iy0[RA], ix0[RA], abase[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + lr + RPP * i;
    if (m < pM) {
      int ox = m % pOWp, r = m / pOWp, oy = r % pOHp, n = r / pOHp;
      iy0[i] = oy * p.in_sy; ix0[i] = ox * p.in_sx; abase[i] = n * p.Hi * p.Wi;
    } else { iy0[i] = -(1 << 20); ix0[i] = 0; abase[i] = 0; }
  }
  if (t < BM) {  // output pixel offset (elements) of every tile row, -1 when out of range
    int m = m0 + t, off = -1;
    if (m < pM) {
      int ox = m % pOWp, r = m / pOWp, oy = r % pOHp, n = r / pOHp;
      off = ((n * p.Ho + oy * p.out_sy + P.out_oy) * p.Wo + ox * p.out_sx + P.out_ox) * p.ldd;
    }
    row_off[t] = off;
  }

  // Register-staged software pipeline over ONE LDS stage (35 KB per block -> 4 blocks = 16 waves per CU): tile kt+1
  // travels HBM -> registers while tile kt is multiplied out of LDS; the other resident blocks cover the rest of the
  // latency.  Measured alternatives (3x3 256->256 @64x64, B=64, fwd): LDS double buffer at 2 blocks/CU 649 TFLOP/s,
  // this form 723, prefetch distance 2 with two register sets (184 VGPRs -> 2 blocks/CU) 616.
  uint4 ra0[RA], rb0[RB];
  const int cmask = (1 << p.cshift) - 1;
  // element offset of every row's window origin: the per-tap part of an A address is then one scalar (dy*Wi+dx)*Ci --
  // no per-row integer multiplies inside the K loop (v_mul_lo_u32 is quarter rate)
  int pix0[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) pix0[i] = iy0[i] >= 0 ? (abase[i] + iy0[i] * p.Wi + ix0[i]) * p.Ci : 0;
  auto tap_of = [&](int kt) -> Tap { return ptaps[((CAT && kt >= nk_main ? nk_main - 1 : kt) * 8) >> p.cshift]; };     // (large-channel path: one tap per K-tile)
  auto load_tile = [&](int kt, const Tap tpu, uint4 (&ra)[RA], uint4 (&rb)[RB]) {
    if constexpr (CAT) {
      if (kt == nk_main) {              // block-uniform: the second operand pair, chunk lc of row m / output channel n
        const bool okc = lc * CH < p.c2;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const int m = m0 + lr + RPP * i;
          ra[i] = buf_load16(rsA2, (okc && m < pM) ? (m * p.c2 + lc * CH) * (int)sizeof(T) : OOB_OFF);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int n = n0 + lr + RPP * i;
          rb[i] = buf_load16(rsB2, (okc && n < p.Nout) ? (n * p.c2 + lc * CH) * (int)sizeof(T) : OOB_OFF);
        }
        return;
      }
    }
    if constexpr (SMALL_C) {
      const int q = kt * 8 + lc; const bool okq = q < pkchunks; const int tap = okq ? (q >> p.cshift) : 0; const int cc = (q & cmask) * CH;
      const Tap tp = ptaps[tap];
      const int koff = (int)tp.widx * p.Ci + cc;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        int iy = iy0[i] + tp.dy, ix = ix0[i] + tp.dx;
        bool ok = okq && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        ra[i] = buf_load16(rsA, ok ? ((abase[i] + iy * p.Wi + ix) * p.Ci + cc) * (int)sizeof(T) : OOB_OFF);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        int n = n0 + lr + RPP * i;
        rb[i] = buf_load16(rsB, (okq && n < p.Nout) ? (n * p.ldb + koff) * (int)sizeof(T) : OOB_OFF);
      }
    } else {
      const int cc = (((kt * 8) & cmask) + lc) * CH;
      const int toff = ((int)tpu.dy * p.Wi + (int)tpu.dx) * p.Ci + cc;
      const int koff = (int)tpu.widx * p.Ci + cc;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int iy = iy0[i] + tpu.dy, ix = ix0[i] + tpu.dx;
        const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        ra[i] = buf_load16(rsA, ok ? (pix0[i] + toff) * (int)sizeof(T) : OOB_OFF);
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        int n = n0 + lr + RPP * i;
        rb[i] = buf_load16(rsB, n < p.Nout ? (n * p.ldb + koff) * (int)sizeof(T) : OOB_OFF);
      }
    }
  };
  // LDS-DMA issue of K-tile kt into ring stage `stage`: lane (row lr, physical chunk lc) fetches the logical chunk
  // lc ^ ((row>>1)&7), so the linear 1-KiB image each wave-instruction writes IS the swizzled tile (swizzle on the source)
  const int lcs = lc ^ ((lr >> 1) & 7);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_tile = [&](int kt, int stage) {
    const bool cat = CAT && kt == nk_main;            // block-uniform
    const int q0 = (cat ? 0 : kt) * 8, tap = q0 >> p.cshift;
    const int cc = ((q0 & cmask) + lcs) * CH;
    const Tap tp = ptaps[tap];
    const int koff = (int)tp.widx * p.Ci + cc;
    const int toff = ((int)tp.dy * p.Wi + (int)tp.dx) * p.Ci + cc;
    (void)koff; (void)toff; (void)wave_u;   // (only used in the device pass below)
#if defined(__HIP_DEVICE_COMPILE__)    // (the host pass must still be able to emit the kernel stub)
    typedef __attribute__((address_space(3))) void* ldsp;
    char* sa = smem + grp * kRing + stage * SM::kStage + wave_u * 1024;
    char* sb = sa + BM * 128;
    if constexpr (CAT) {
      if (cat) {                                      // logical chunk lcs of row m / output channel n of the second operand pair
        const bool okc = lcs * CH < p.c2;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          const int m = m0 + lr + RPP * i;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (ldsp)(sa + i * RPP * 128), 16,
                                                   (okc && m < pM) ? (m * p.c2 + lcs * CH) * (int)sizeof(T) : OOB_OFF, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const int n = n0 + lr + RPP * i;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (ldsp)(sb + i * RPP * 128), 16,
                                                   (okc && n < p.Nout) ? (n * p.c2 + lcs * CH) * (int)sizeof(T) : OOB_OFF, 0, 0, 0);
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int iy = iy0[i] + tp.dy, ix = ix0[i] + tp.dx;
      bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsp)(sa + i * RPP * 128), 16,
                                               ok ? (pix0[i] + toff) * (int)sizeof(T) : OOB_OFF, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int n = n0 + lr + RPP * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsp)(sb + i * RPP * 128), 16,
                                               n < p.Nout ? (n * p.ldb + koff) * (int)sizeof(T) : OOB_OFF, 0, 0, 0);
    }
#endif
  };
  char* as = smem;
  char* bs = smem + SM::kARows * 128;
  auto store_tile = [&](const uint4 (&ra)[RA], const uint4 (&rb)[RB]) {
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<uint4*>(as + swz128(lr + RPP * i, lc)) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<uint4*>(bs + swz128(lr + RPP * i, lc)) = rb[i];
  };

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int r31 = lane & 31, hi = lane >> 5;
  auto compute = [&]() {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8_t a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(as + swz128(wm0 + i * 32 + r31, 2 * s + hi));
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(bs + swz128(wn0 + j * 32 + r31, 2 * s + hi));
        // operands swapped: acc holds D^T (lane = m row, registers = 4-wide runs of n) for a wide epilogue
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int e = 2 * s + hi;
        float a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float*>(as + swz128(wm0 + i * 32 + r31, e >> 2) + (e & 3) * 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float*>(bs + swz128(wn0 + j * 32 + r31, e >> 2) + (e & 3) * 4);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    }
  };

  const int nk = nk_main + (CAT ? 1 : 0);
  if constexpr (KW3) {
    // sub-step ks = (group g, kw), group g = (kernel row kh, channel chunk): tap index kh*3 + kw, channels chunk*64 ..
    const int nchunk = p.Ci >> 6, nsub = 9 * nchunk;
    for (int i = t; i < SM::kARows * 8; i += NTHR) reinterpret_cast<uint4*>(as)[i] = make_uint4(0, 0, 0, 0);
    int simg[RA];                                         // LDS byte offset of this thread's A rows in the segmented image
#pragma unroll
    for (int i = 0; i < RA; ++i) { const int r = lr + RPP * i; simg[i] = swz128(r + 2 + 4 * (r >> p.lw), lc); }
    int rimg[MT];                                         // image row of this lane's fragment rows
#pragma unroll
    for (int i = 0; i < MT; ++i) { const int q = wm0 + i * 32 + r31; rimg[i] = q + 2 + 4 * (q >> p.lw); }
    auto load_a = [&](int g, uint4 (&ra)[RA]) {
      const int kh = g / nchunk, chunk = g - kh * nchunk;
      const int dyv = ptaps[kh * 3].dy;
      const int toff = dyv * p.Wi * p.Ci + chunk * 64 + lc * CH;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const bool ok = (unsigned)(iy0[i] + dyv) < (unsigned)p.Hi;
        ra[i] = buf_load16(rsA, ok ? (pix0[i] + toff) * (int)sizeof(T) : OOB_OFF);
      }
    };
    auto load_b = [&](int ks, uint4 (&rb)[RB]) {
      const int g = ks / 3, kw = ks - g * 3;
      const int kh = g / nchunk, chunk = g - kh * nchunk;
      const int koff = (int)ptaps[kh * 3 + kw].widx * p.Ci + chunk * 64 + lc * CH;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int n = n0 + lr + RPP * i;
        rb[i] = buf_load16(rsB, n < p.Nout ? (n * p.ldb + koff) * (int)sizeof(T) : OOB_OFF);
      }
    };
    auto compute_kw = [&](int dxv) {
      int arow[MT], axor[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) { const int r = rimg[i] + dxv; arow[i] = r * 128; axor[i] = (r >> 1) & 7; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8_t a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(as + arow[i] + (((2 * s + hi) ^ axor[i]) << 4));
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(bs + swz128(wn0 + j * 32 + r31, 2 * s + hi));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    };
    load_a(0, ra0);
    load_b(0, rb0);
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
        load_b(ks + 1, rb0);
        if (kw == 2) load_a(g + 1, ra0);                  // the next group's A tile travels during this group's last sub-step
      }
      __builtin_amdgcn_s_setprio(1);
      compute_kw(dxv);
      __builtin_amdgcn_s_setprio(0);
      if (++kw == 3) { kw = 0; ++g; }
    }
  } else if constexpr (DMA && KG > 1) {
    // K group g multiplies K tiles [kbeg, kend) out of its own ring; the barriers are the block's, so both groups make nk0 trips
    const int nk0 = (nk + 1) >> 1;
    const int kbeg = grp ? nk0 : 0, kend = grp ? nk : nk0;
    char* ring = smem + grp * kRing;
    if (kbeg < kend) dma_tile(kbeg, 0);
    for (int i = 0; i < nk0; ++i) {
      const int kt = kbeg + i;
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      __syncthreads();
      if (kt + 1 < kend) dma_tile(kt + 1, (i + 1) & 1);
      as = ring + (i & 1) * SM::kStage; bs = as + BM * 128;
      if (kt < kend) {
        __builtin_amdgcn_s_setprio(1);
        compute();
        __builtin_amdgcn_s_setprio(0);
      }
    }
    // group 1's partial sums travel to group 0 through group 1's ring space: [wave][accumulator quad][lane] float4, written and
    // read by the same (wave, lane) pair of each group -- conflict-free, no index arithmetic beyond a constant stride
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem + kRing) + wave * (MT * NT * 4 * 64) + lane;
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            red[((i * NT + j) * 4 + q) * 64] = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = red[((i * NT + j) * 4 + q) * 64];
            acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
          }
    }
  } else if constexpr (DMA) {
    dma_tile(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile kt have landed in LDS
#endif
      __syncthreads();                                      // everyone's have; stage (kt+1)&1 is no longer being read
      if (kt + 1 < nk) dma_tile(kt + 1, (kt + 1) & 1);
      as = smem + (kt & 1) * SM::kStage; bs = as + BM * 128;
      __builtin_amdgcn_s_setprio(1);
      compute();
      __builtin_amdgcn_s_setprio(0);
    }
  } else {
    // (two K-tiles in flight for the small tiles -- 64x128 and below have the registers -- measured: no gain on the
    //  latency-bound mid-size layers, occupancy 5 -> 3; per-K-step issue cost, not prefetch depth, bounds them)
    load_tile(0, tap_of(0), ra0, rb0);
    Tap tpn = tap_of(nk > 1 ? 1 : 0);                      // the tap word of tile kt+1 is fetched one iteration early
    for (int kt = 0; kt < nk; ++kt) {
      const Tap tpu = tpn;
      if (kt + 2 < nk) tpn = tap_of(kt + 2);
      __syncthreads(); store_tile(ra0, rb0); __syncthreads();
      if (kt + 1 < nk) load_tile(kt + 1, tpu, ra0, rb0);
      __builtin_amdgcn_s_setprio(1);
      compute();
      __builtin_amdgcn_s_setprio(0);
    }
  }
  __syncthreads();

  char* outs = smem;
  if constexpr (HM_OUT) {
    // ---- heat-map epilogue: acc + bias -> fp32 LDS tile -> per key-point runs of BM consecutive pixels (512-byte rows)
    constexpr int HS = BN * 4 + 16;
    static_assert(BM * HS <= SM::kBytes - BM * 4, "fp32 staging tile must fit");
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ml = wm0 + i * 32 + r31, nl = wn0 + j * 32 + 8 * g + 4 * hi;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + ((p.bias && (nl + e) < p.Nout) ? p.bias[nl + e] : 0.f);
          *reinterpret_cast<float4*>(outs + ml * HS + nl * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    __syncthreads();
    float* __restrict__ Y = reinterpret_cast<float*>(p.D);
    for (int id = t; id < p.Nout * BM; id += NTHR) {
      const int k = id / BM, r = id % BM;
      const int gp = row_off[r];                       // global pixel index (ldd == 1)
      if (gp < 0) continue;
      const int img = gp / p.hw, pp = gp - img * p.hw;
      Y[((size_t)img * p.Nout + k) * p.hw + pp] = *reinterpret_cast<const float*>(outs + r * HS + k * 4);
    }
    return;
  }
  // ---- epilogue: (acc + bias) * scale -> T -> LDS tile -> coalesced 16-byte rows (+residual / +dx)
  const float scale = p.scale ? *p.scale : 1.0f;
  if (KG == 1 || grp == 0)       // (KG > 1: K group 0 holds the sums; every thread of the block then streams rows out)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ml = wm0 + i * 32 + r31;
        const int nl = wn0 + j * 32 + 8 * g + 4 * hi;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float b = (p.bias && (n0 + nl + e) < p.Nout) ? p.bias[n0 + nl + e] : 0.f;
          if constexpr (CAT) { if (p.bias2 && (n0 + nl + e) < p.Nout) b += p.bias2[n0 + nl + e]; }
          v[e] = (acc[i][j][4 * g + e] + b) * scale;
        }
        char* dst = outs + ml * SM::kOutStride + nl * (int)sizeof(T);
        if constexpr (sizeof(T) == 2) {
          union { bf16_t h[4]; uint2 q; } u;
#pragma unroll
          for (int e = 0; e < 4; ++e) u.h[e] = (bf16_t)v[e];
          *reinterpret_cast<uint2*>(dst) = u.q;
        } else {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
  __syncthreads();
  constexpr int CPR = BN / CH;  // 16-byte chunks per tile row
  T* __restrict__ D = reinterpret_cast<T*>(p.D);
  const T* __restrict__ R = reinterpret_cast<const T*>(p.residual);
  // Optional fused BatchNorm statistics of the tile just produced (per output channel over the tile's valid rows, Welford):
  // every thread already holds the rounded values it streams out, so the statistics cost no extra LDS or HBM reads.
  constexpr bool stats = EPI == 1, bnb = EPI == 2;
  float sn = 0.f, smean[CH], sm2[CH];      // bnb mode reuses smean / sm2 as the two running sums
  float bmu[CH], bis[CH], bsc[CH], bsh[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { smean[e] = 0.f; sm2[e] = 0.f; bmu[e] = 0.f; bis[e] = 0.f; bsc[e] = 0.f; bsh[e] = 0.f; }
  const T* __restrict__ BX = reinterpret_cast<const T*>(p.bnb_x);
  const T* __restrict__ BY = reinterpret_cast<const T*>(p.bnb_y);
  if (bnb) {
    const int n = n0 + (t % CPR) * CH;
    if (n < p.Nout) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        bmu[e] = p.bnb_mean[n + e]; bis[e] = p.bnb_invstd[n + e];
        if (p.bnb_relu == 2) { bsc[e] = p.bnb_gamma[n + e] * bis[e]; bsh[e] = p.bnb_beta[n + e] - bmu[e] * bsc[e]; }
      }
    }
  }
  if constexpr (!bnb) {
    // rows streamed one at a time, loads interleaved with the stores (measured faster than fetching all rows' addends
    // up front: 44 vs 49 us on the 256->64 1x1 dgrad @64x64)
    for (int id = t; id < BM * CPR; id += NTHR) {
      const int r = id / CPR, c = id % CPR;
      const int off = row_off[r];
      const int n = n0 + c * CH;
      if (off < 0 || n >= p.Nout) continue;
      float v[CH];
      Chunk<T>::load(reinterpret_cast<const T*>(outs + r * SM::kOutStride + c * 16), v);
      const size_t g = (size_t)off + n;
      if (stats) {
        // (Welford per row.  A cheaper form -- running sums of deviations from the thread's first row, 3 VALU per element instead
        //  of 6 + a division per row -- is as accurate (1e-8 of float64, both) and halves this epilogue's VALU work in isolation,
        //  but measured no gain in the iteration: 33.27 / 33.34 vs 33.26 / 33.13 ms, same box; other resident blocks hide it)
        sn += 1.f; const float inv = 1.f / sn;
#pragma unroll
        for (int e = 0; e < CH; ++e) { const float d = v[e] - smean[e]; smean[e] += d * inv; sm2[e] += d * (v[e] - smean[e]); }
      }
      if (R) { float w[CH]; Chunk<T>::load(R + g, w);
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] += w[e]; }
      if (p.accumulate) {
        // the optional bit mask is applied to the packed words BEFORE unpacking: selecting per unpacked float made the
        // compiler permute the registers and feed v_pk_add_f32 through op_sel, and those adds dropped an addend now and
        // then (lanes 48-63 only, run-to-run different: 40 - 240 of 8.4 M elements on the 32x64x64x64 dgrad) on gfx950
        uint4 q = *reinterpret_cast<const uint4*>(D + g);
        if (p.acc_mask) q = keep_masked<T>(q, p.acc_mask[g / CH]);
        float w[CH]; Chunk<T>::unpack(q, w);
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] += w[e]; }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] = v[e] < 0.f ? 0.f : v[e]; }      // (keeps NaN, like ATen relu)
      Chunk<T>::store(D + g, v);
    }
  } else {
    // EPI == 2 (own register budget, 2 blocks per CU): every global operand of the epilogue -- the addend and the
    // BatchNorm's x / y -- is fetched for all of this thread's rows before any is used: one latency instead of one per row
    constexpr int IT = BM * CPR / NTHR;
    static_assert(BM * CPR % NTHR == 0, "whole passes only");
    const int ec = t % CPR, er0 = t / CPR;
    const int en = n0 + ec * CH;
    const bool ecok = en < p.Nout;
    int eoff[IT];
    uint4 qa[IT], qx[IT], qy[IT];
    const T* __restrict__ ADD = R ? R : (p.accumulate ? D : nullptr);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int off = row_off[er0 + k * (NTHR / CPR)];
      eoff[k] = (off >= 0 && ecok) ? off + en : -1;
      if (eoff[k] >= 0) {
        if (ADD) qa[k] = *reinterpret_cast<const uint4*>(ADD + (size_t)eoff[k]);
        qx[k] = *reinterpret_cast<const uint4*>(BX + (size_t)eoff[k]);
        if (p.bnb_relu == 1) qy[k] = *reinterpret_cast<const uint4*>(BY + (size_t)eoff[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      if (eoff[k] < 0) continue;
      const int r = er0 + k * (NTHR / CPR);
      float v[CH];
      Chunk<T>::load(reinterpret_cast<const T*>(outs + r * SM::kOutStride + ec * 16), v);
      const size_t g = (size_t)eoff[k];
      if (ADD) { float w[CH]; Chunk<T>::unpack(qa[k], w);
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] += w[e]; }
      if (R && p.accumulate) { float w[CH]; Chunk<T>::load(D + g, w);      // (both at once: not used by this model)
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] += w[e]; }
      Chunk<T>::store(D + g, v);
      // same arithmetic as bn_bwd_reduce_kernel, on the value as stored (rounded to T)
      float xv[CH], gq[CH]; Chunk<T>::unpack(qx[k], xv);
#pragma unroll
      for (int e = 0; e < CH; ++e) gq[e] = (float)(T)v[e];
      if (p.bnb_relu == 1) { float yv[CH]; Chunk<T>::unpack(qy[k], yv);
#pragma unroll
        for (int e = 0; e < CH; ++e) gq[e] = yv[e] > 0.f ? gq[e] : 0.f; }
      else if (p.bnb_relu == 2) {
#pragma unroll
        for (int e = 0; e < CH; ++e) gq[e] = (xv[e] * bsc[e] + bsh[e]) > 0.f ? gq[e] : 0.f; }
#pragma unroll
      for (int e = 0; e < CH; ++e) { smean[e] += gq[e]; sm2[e] += gq[e] * ((xv[e] - bmu[e]) * bis[e]); }
    }
  }
  if (bnb) {
    constexpr int NW = NTHR / 64;
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < CH; ++e) { smean[e] += __shfl_down(smean[e], o, 64); sm2[e] += __shfl_down(sm2[e], o, 64); }
    }
    __syncthreads();
    float* sp = reinterpret_cast<float*>(smem);        // [NW][BN][2]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        float* q = sp + ((size_t)wave_blk * BN + (t % CPR) * CH + e) * 2;
        q[0] = smean[e]; q[1] = sm2[e];
      }
    }
    __syncthreads();
    if (t < BN && n0 + t < p.Nout) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a1 += sp[((size_t)w * BN + t) * 2]; a2 += sp[((size_t)w * BN + t) * 2 + 1]; }
      const int slice = (tile / p.ntn) * p.nphase + phi;
      float* out = p.bnb_partial + ((size_t)slice * p.Nout + n0 + t) * 2;
      out[0] = a1; out[1] = a2;
    }
    return;
  }
  if (stats) {
    // thread t owns chunk column t % CPR and the rows t / CPR + k * (NTHR / CPR).  Fold the row lanes: inside a wave by
    // shuffle-down (lower lane = left operand), across the waves through LDS in wave order (fixed order: reproducible).
    constexpr int NW = NTHR / 64;
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1) {
      const float nb = __shfl_down(sn, o, 64);
      const float nt = sn + nb, f = nt > 0.f ? nb / nt : 0.f;
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const float mb = __shfl_down(smean[e], o, 64), vb = __shfl_down(sm2[e], o, 64);
        const float d = mb - smean[e];
        smean[e] += d * f; sm2[e] += vb + d * d * sn * f;
      }
      sn = nt;
    }
    __syncthreads();                                   // everyone is done reading the staged tile
    float* sp = reinterpret_cast<float*>(smem);        // [NW][BN][3]
    static_assert(NW * BN * 3 * 4 <= SM::kBytes - BM * 4, "statistics scratch must fit in the tile staging area");
    if (lane < CPR && lane < 64) {
      const int c = t % CPR;
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        float* q = sp + ((size_t)wave_blk * BN + c * CH + e) * 3;
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

// ------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
  const void* X; const void* DY; float* out;
  int Hi, Wi, Ci, Ho, Wo, Co;
  int kw, stride, pad, cshift;
  int M, rows_per_split, ldw;
  long slab_stride;
  int nto, nti;
  unsigned x_bytes, dy_bytes;
  FastDiv dWo, dHo;
};

// tr16-read friendly swizzle of a [rows][256 B] bf16 image (guide T10, image (b))
__device__ __forceinline__ int swz256(int r, int ch) { return r * 256 + ((ch ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 4); }

template <typename T> struct WgradSmem {
  static constexpr int BKM = (sizeof(T) == 2) ? 64 : 32;
  static constexpr int kBytes = 2 * BKM * 128 * (int)sizeof(T) + 2 * BKM * 8;
};
// body shared by the single-problem kernel and the grouped one (`bid` of `nblk` blocks work on problem p)
template <typename T>
__device__ __forceinline__ void wgrad_gemm_body(const WgradArgs& p, const int bid, const int nblk, char* smem) {
  constexpr int CH = MmaTraits<T>::CH;
  constexpr int BKM = (sizeof(T) == 2) ? 64 : 32;   // reduction rows per LDS tile
  constexpr int CPR = 128 / CH;                      // chunks per 128-wide tile row
  constexpr int RPT = BKM * CPR / 256;               // rows per thread per operand
  constexpr int ROWB = 128 * (int)sizeof(T);         // LDS row bytes
  char* ys = smem;                 // DY tile [BKM][128 o]
  char* xs = smem + BKM * ROWB;    // X  tile [BKM][128 cols]
  // pixel decode of the reduction rows, shared by the 16/32 lanes that stage one row: one thread per row decodes
  // (n, oy, ox) for the NEXT tile while this one is multiplied -> {pixel index of the window origin, (iy0<<16)|ix0}
  int2* rowinfo = reinterpret_cast<int2*>(smem + 2 * BKM * ROWB);   // [2][BKM]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // XCD-aware order: all tiles of one pixel range (split) are neighbours in the remapped id, i.e. share an XCD / L2,
  // so the DY and X rows of that range are fetched into one L2 instead of eight
  const int ntile = p.nto * p.nti;
  const int lid = xcd_remap(bid, nblk);
  const int split = lid / ntile, tile = lid - split * ntile;
  const int o0 = (tile / p.nti) * 128, c0 = (tile % p.nti) * 128;  // c0: flattened (tap, ci) column
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int lc = t % CPR, lr = t / CPR;

  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes), rsY = make_rsrc(p.DY, p.dy_bytes);

  // this thread's fixed column chunk -> (tap, ci)
  const int col = c0 + lc * CH;
  const bool colok = col < p.ldw;
  const int qc = col / CH;
  const int tap = colok ? (qc >> p.cshift) : 0;
  const int ci = (qc & ((1 << p.cshift) - 1)) * CH;
  const int tdy = tap / p.kw - p.pad, tdx = tap % p.kw - p.pad;
  const bool ook = (o0 + lc * CH) < p.Co;

  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);

  uint4 rx[RPT], ry[RPT];
  const int toff = tdy * p.Wi + tdx;                       // this thread's tap as a pixel-index offset
  auto decode_rows = [&](int mt0, int buf) {               // threads 0..BKM-1, one reduction row each
    if (t < BKM) {
      const unsigned m = (unsigned)(mt0 + t);
      unsigned r = fd_div(m, p.dWo); int ox = (int)(m - r * p.Wo);
      unsigned n = fd_div(r, p.dHo); int oy = (int)(r - n * p.Ho);
      const int iy0 = oy * p.stride, ix0 = ox * p.stride;
      rowinfo[buf * BKM + t] = make_int2(((int)n * p.Hi + iy0) * p.Wi + ix0, (iy0 << 16) | ix0);
    }
  };
  auto load_tile = [&](int mt0, int buf) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int rr = lr + (256 / CPR) * i;
      const int m = mt0 + rr;
      const bool mok = m < mend;
      ry[i] = buf_load16(rsY, (mok && ook) ? (m * p.Co + o0 + lc * CH) * (int)sizeof(T) : OOB_OFF);
      const int2 ri = rowinfo[buf * BKM + rr];
      const int iy = (ri.y >> 16) + tdy, ix = (ri.y & 0xffff) + tdx;
      const bool xok = mok && colok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      rx[i] = buf_load16(rsX, xok ? ((ri.x + toff) * p.Ci + ci) * (int)sizeof(T) : OOB_OFF);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = lr + (256 / CPR) * i;
      int off;
      if constexpr (sizeof(T) == 2) off = swz256(r, lc); else off = r * ROWB + lc * 16;
      *reinterpret_cast<uint4*>(ys + off) = ry[i];
      *reinterpret_cast<uint4*>(xs + off) = rx[i];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int r31 = lane & 31, hi = lane >> 5;
  // tr16 lane roles: 16-lane group g reads a 4-row x 16-col block; lane 4q+p supplies row q, cols 4p..4p+3
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  int trA[2][2], trB[2][2];     // LDS byte addresses of the tr16 reads of k-step 0: [sub-tile i][rows +0 / +4]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 8 * (tg >> 1) + tq + 4 * u;
      trA[i][u] = swz256(row, (wm0 + i * 32) / 8 + 2 * (tg & 1) + (tp >> 1)) + 8 * (tp & 1);
      trB[i][u] = swz256(row, (wn0 + i * 32) / 8 + 2 * (tg & 1) + (tp >> 1)) + 8 * (tp & 1);
    }
  decode_rows(mbeg, 0);
  __syncthreads();
  if (mbeg < mend) load_tile(mbeg, 0);
  int buf = 0;
  for (int mt0 = mbeg; mt0 < mend; mt0 += BKM, buf ^= 1) {
    __syncthreads();   // previous tile's LDS reads done
    store_tile();
    decode_rows(mt0 + BKM, buf ^ 1);
    __syncthreads();
    if (mt0 + BKM < mend) load_tile(mt0 + BKM, buf ^ 1);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < BKM / 16; ++s) {
        bf16x8_t a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          typedef __attribute__((address_space(3))) bf16x4_t* lds4;
          // the swizzle term of swz256 does not depend on s (16 rows = 4096 B per k-step): base + immediate offset
          bf16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(ys + trA[i][0] + s * 4096));
          bf16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(ys + trA[i][1] + s * 4096));
          bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(xs + trB[i][0] + s * 4096));
          bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(xs + trB[i][1] + s * 4096));
          a[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
          b[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < BKM / 2; ++s) {
        const int row = 2 * s + hi;
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const float*>(ys + row * ROWB + (wm0 + i * 32 + r31) * 4);
          b[i] = *reinterpret_cast<const float*>(xs + row * ROWB + (wn0 + i * 32 + r31) * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int c = c0 + wn0 + j * 32 + r31;
        if (o < p.Co && c < p.ldw) out[(size_t)o * p.ldw + c] = acc[i][j][r];
      }
}

template <typename T>
__global__ __launch_bounds__(256, 3) void wgrad_gemm_kernel(const WgradArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[WgradSmem<T>::kBytes];
  wgrad_gemm_body<T>(p, blockIdx.x, gridDim.x, smem);
}

// Several independent weight-gradient problems in ONE launch (the small layers of a ResNet stage: each alone offers
// 16 .. 64 output tiles, far too few for 256 CUs, so the single-problem launch splits its pixel range 16 .. 48 ways and
// pays for it with as many fp32 slabs plus a reduce launch).  Grouped, the tiles of all problems fill the chip together:
// 2 .. 4 splits per problem, long K loops, one launch + one grouped slab reduce for the whole stage.  The argument blocks
// travel by value in the kernel-argument segment (graph-capturable, no device table).
#define WG_MAX 24
struct WgradGroupArgs { WgradArgs p[WG_MAX]; int bstart[WG_MAX + 1]; int n; };
template <typename T>
__global__ __launch_bounds__(256, 3) void wgrad_group_kernel(const WgradGroupArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[WgradSmem<T>::kBytes];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.bstart[i + 1]) ++i;         // block-uniform scan of <= 24 entries
  wgrad_gemm_body<T>(g.p[i], (int)blockIdx.x - g.bstart[i], g.bstart[i + 1] - g.bstart[i], smem);
}

// 256 x 256 output tiles for the grouped 1x1 weight gradients with >= 256 channels on both sides (bf16; layer3 / layer4 of the
// ResNets).  By counters the 128 x 128 form fetches 2 - 7x its operands from beyond L2 (the sibling tiles of a pixel range do not find
// each other's lines in the 4-MB L2s: the layer4 group moved 0.94 GB through L1 for 0.15 GB of x + dy), i.e. those launches are
// bound by L2-miss bandwidth.  Here a block of 8 waves owns 256 output channels x 256 input channels: per 64-pixel step four
// [64][128] images (two DY halves, two X halves; same tr16 swizzle and fragment reads as the 128 x 128 body) = 64 KB for 256 MFMAs
// -- half the bytes per MFMA -- travel global -> LDS by LDS-DMA into a two-stage ring (the swizzle applied on the per-lane source
// address, out-of-range lanes deliver zeros; one barrier per step, no staging registers: a register-staged form of this tile, one
// block per CU with two barriers per step, measured SLOWER than the 128 x 128 kernel: 265 vs 223 us on the layer3 group), and every
// wave multiplies a 128 x 64 sub-tile (128 accumulators).  1x1 convs only (stride 1 or 2): a column is an input channel.
#define WG256_IMG (64 * 256)
template <int OFF> __device__ __forceinline__ bf16x4_t wg256_tr(unsigned a) {
  bf16x4_t v; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}
// one 16-pixel sub-step of a wave's 128 x 64 sub-tile: 12 transposed fragment reads (asm), one counted wait that names them, 8 MFMAs
template <int OFF> __device__ __forceinline__ void wg256_step(unsigned st, const int (&trA)[4][2], const int (&trB)[2][2], f32x16_t (&acc)[4][2]) {
  bf16x4_t a0[4], a1[4], b0[2], b1[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a0[i] = wg256_tr<OFF>(st + (unsigned)trA[i][0]); a1[i] = wg256_tr<OFF>(st + (unsigned)trA[i][1]); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { b0[j] = wg256_tr<OFF>(st + (unsigned)trB[j][0]); b1[j] = wg256_tr<OFF>(st + (unsigned)trB[j][1]); }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]),
               "+v"(b0[0]), "+v"(b0[1]), "+v"(b1[0]), "+v"(b1[1]));
  bf16x8_t a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = __builtin_shufflevector(a0[i], a1[i], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
  for (int j = 0; j < 2; ++j) b[j] = __builtin_shufflevector(b0[j], b1[j], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
}
#define WG256_SMEM (2 * 4 * WG256_IMG + 2 * 64 * 4)
__device__ __forceinline__ void wgrad_gemm256_body(const WgradArgs& p, const int bid, const int nblk, char* smem) {
  constexpr int BKM = 64, IMG = WG256_IMG, STAGE = 4 * IMG;
  int* rowpix = reinterpret_cast<int*>(smem + 2 * STAGE);      // [2][BKM]: input pixel index of every reduction row
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntile = p.nto * p.nti;                   // (in 256 x 256 tiles)
  const int lid = xcd_remap(bid, nblk);
  const int split = lid / ntile, tile = lid - split * ntile;
  const int o0 = (tile / p.nti) * 256, c0 = (tile % p.nti) * 256;
  const int wo = wave >> 2, wc = wave & 3;           // multiply: DY image wo (128 output channels), X image wc >> 1, columns (wc & 1) * 64 ..
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes), rsY = make_rsrc(p.DY, p.dy_bytes);
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);
  // staging: wave w fills image w >> 1 (0, 1: DY halves, 2, 3: X halves), row groups (w & 1) * 8 + j, j = 0 .. 7, of four rows each;
  // lane -> row (lane >> 4) of the group, physical chunk lane & 15, which holds the logical chunk pc ^ (((r & 3) << 2) | ((r >> 2) & 3))
  const int img = wave >> 1, half = img & 1;
  const bool isx = img >= 2;
  const int rsub = lane >> 4, pc = lane & 15;
  int colb[4];                                       // byte offset of this lane's column for row groups with (group & 3) == q
#pragma unroll
  for (int q = 0; q < 4; ++q) colb[q] = ((isx ? c0 : o0) + half * 128 + ((pc ^ ((rsub << 2) | q)) << 3)) * 2;
  const int ld2 = (isx ? p.Ci : p.Co) * 2;           // row pitch in bytes
  auto decode_rows = [&](int mt0, int buf) {
    if (t < BKM) {
      const unsigned m = (unsigned)(mt0 + t);
      unsigned r = fd_div(m, p.dWo); int ox = (int)(m - r * p.Wo);
      unsigned n = fd_div(r, p.dHo); int oy = (int)(r - n * p.Ho);
      rowpix[buf * BKM + t] = ((int)n * p.Hi + oy * p.stride) * p.Wi + ox * p.stride;
    }
  };
  auto dma_tile = [&](int mt0, int stage, int buf) {
    (void)mt0; (void)stage; (void)buf;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* ldsp;
    char* dst = smem + stage * STAGE + img * IMG + (wave & 1) * 8 * 1024;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = ((wave & 1) * 8 + j) * 4 + rsub, m = mt0 + r;
      const int row = isx ? rowpix[buf * BKM + r] : m;
      const int off = m < mend ? row * ld2 + colb[j & 3] : OOB_OFF;
      if (isx) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (ldsp)(dst + j * 1024), 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (ldsp)(dst + j * 1024), 16, off, 0, 0, 0);
    }
#endif
  };
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int r31 = lane & 31, hi = lane >> 5;
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;      // tr16 lane roles (as in wgrad_gemm_body)
  const int wn0 = (wc & 1) * 64;
  int trA[4][2], trB[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 8 * (tg >> 1) + tq + 4 * u;
#pragma unroll
    for (int i = 0; i < 4; ++i) trA[i][u] = wo * IMG + swz256(row, (i * 32) / 8 + 2 * (tg & 1) + (tp >> 1)) + 8 * (tp & 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) trB[j][u] = (2 + (wc >> 1)) * IMG + swz256(row, (wn0 + j * 32) / 8 + 2 * (tg & 1) + (tp >> 1)) + 8 * (tp & 1);
  }
  typedef __attribute__((address_space(3))) char* lds_cptr256;
  const unsigned smem_lds = (unsigned)(uintptr_t)(lds_cptr256)smem;
  decode_rows(mbeg, 0);
  decode_rows(mbeg + BKM, 1);
  __syncthreads();
  if (mbeg < mend) dma_tile(mbeg, 0, 0);
  int it = 0;
  for (int mt0 = mbeg; mt0 < mend; mt0 += BKM, ++it) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of tile `it` have landed
#endif
    __syncthreads();                                         // everyone's have; stage (it + 1) & 1 is no longer being read
    if (mt0 + BKM < mend) dma_tile(mt0 + BKM, (it + 1) & 1, (it + 1) & 1);
    decode_rows(mt0 + 2 * BKM, it & 1);                      // (row buffer it & 1 was consumed by the DMA issued one trip ago)
    // The fragment reads are inline asm: hipcc guards every LDS read it can see behind an LDS-DMA with `s_waitcnt vmcnt(0)`, i.e. it
    // would wait for the tile issued a few lines up before multiplying this one (seen in the disassembly of the builtin form).
    const unsigned st = smem_lds + (unsigned)((it & 1) * STAGE);
    __builtin_amdgcn_s_setprio(1);
    wg256_step<0>(st, trA, trB, acc); wg256_step<4096>(st, trA, trB, acc); wg256_step<8192>(st, trA, trB, acc); wg256_step<12288>(st, trA, trB, acc);
    __builtin_amdgcn_s_setprio(0);
  }
  float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wo * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int c = c0 + (wc >> 1) * 128 + wn0 + j * 32 + r31;
        if (o < p.Co && c < p.ldw) out[(size_t)o * p.ldw + c] = acc[i][j][r];
      }
}
__global__ __launch_bounds__(512, 1) void wgrad_group256_kernel(const WgradGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem256[];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.bstart[i + 1]) ++i;
  wgrad_gemm256_body(g.p[i], (int)blockIdx.x - g.bstart[i], g.bstart[i + 1] - g.bstart[i], smem256);
}

// grouped fixed-order slab reduction: out_i (=|+=) sum over the S_i slabs of problem i, one float4 column per thread
struct SlabItem { const float* slabs; float* out; long n4, stride; int S, accumulate; };
struct SlabGroupArgs { SlabItem it[WG_MAX]; int bstart[WG_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void slab_reduce_group_kernel(const SlabGroupArgs g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.bstart[i + 1]) ++i;
  const SlabItem& q = g.it[i];
  const long c = (long)((int)blockIdx.x - g.bstart[i]) * 256 + threadIdx.x;
  if (c >= q.n4) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < q.S; ++s) {
    const float4 a = reinterpret_cast<const float4*>(q.slabs + (size_t)s * q.stride)[c];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
  }
  if (q.accumulate) { const float4 o = reinterpret_cast<const float4*>(q.out)[c]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
  reinterpret_cast<float4*>(q.out)[c] = v;
}

// ------------------------------------------------------------------------------------ wgrad, 3x3 stride 1 (bf16)
// The three horizontal taps of one kernel row share ONE staged X tile: with unit stride the tap (kh, kw) operand of
// output pixel m is the input pixel m + (kh-1)*W + (kw-1), i.e. the kw taps are the same rows shifted by -1 / 0 / +1.
// A block owns (128|64 output channels) x (64 input channels) x (kh; kw = 0,1,2): per 64-pixel step it stages one DY
// tile and one X tile (the X tile once instead of three times) and issues 3x the MFMAs of the generic kernel's step.
// Row shifts must not leak across image rows: the LDS X image keeps every run of min(W,64) pixels in its own segment
// with 4 spare rows between segments (the row before / after a segment is the halo: zero at the image border, the
// neighbouring pixel when W > 64), so a shifted tr16 read picks up zeros exactly where the padding is.
struct WgradKwArgs {
  const void* X; const void* DY; float* out;
  int H, W, Ci, Co;
  int lw, lwf, halo;            // log2(min(W,64)), log2(W), W > 64
  int M, rows_per_split, ldw;
  long slab_stride;
  int nto, nci;
  unsigned x_bytes, dy_bytes;
  FastDiv dH;
};
// 128-byte rows, tr16-read friendly for any 4 consecutive rows (shifted reads included)
__device__ __forceinline__ int swzx(int r, int ch) { return r * 128 + ((ch ^ (((r >> 1) & 1) << 2)) << 4); }

template <int MT> struct WgradKwSmem { static constexpr int kBytes = 64 * 256 + (64 + 4 * 8 + 4) * 128 + 2 * 64 * 4; };
// body shared by the single-problem kernel and the grouped one (`bid` of `nblk` blocks work on problem p)
template <int MT>
__device__ __forceinline__ void wgrad_kw_body(const WgradKwArgs& p, const int bid, const int nblk, char* smem) {
  constexpr int BO = 64 * MT, BKM = 64;
  constexpr int CPRY = BO / 8, RPY = 256 / CPRY, NPY = BKM / RPY;
  constexpr int XROWS = 64 + 4 * 8 + 4;
  char* ys = smem;
  char* xs = smem + BKM * 256;
  int* rowinfo = reinterpret_cast<int*>(smem + BKM * 256 + XROWS * 128);   // [2][BKM]: image row oy of each pixel

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ntile = p.nto * 3 * p.nci;
  const int lid = xcd_remap(bid, nblk);
  const int split = lid / ntile;
  int tile = lid - split * ntile;
  const int ot = tile / (3 * p.nci); tile -= ot * 3 * p.nci;
  const int kh = tile / p.nci, cit = tile - kh * p.nci;
  const int o0 = ot * BO, ci0 = cit * 64, tdy = kh - 1;
  const int wm0 = (wave >> 1) * (32 * MT), wn0 = (wave & 1) * 32;

  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes), rsY = make_rsrc(p.DY, p.dy_bytes);
  const int lcy = t % CPRY, lry = t / CPRY;
  const int lcx = t & 7, lrx = t >> 3;
  const bool ook = (o0 + lcy * 8) < p.Co;
  const bool cok = (ci0 + lcx * 8) < p.Ci;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);

  for (int i = t; i < XROWS * 8; i += 256) reinterpret_cast<uint4*>(xs)[i] = make_uint4(0, 0, 0, 0);

  uint4 ry[NPY], rx[2], rh = make_uint4(0, 0, 0, 0);
  auto decode_rows = [&](int mt0, int buf) {
    if (t < BKM) {
      const unsigned m = (unsigned)(mt0 + t);
      const unsigned r = m >> p.lwf;
      const unsigned n = fd_div(r, p.dH);
      rowinfo[buf * BKM + t] = (m < (unsigned)p.M) ? (int)(r - n * p.H) : -(1 << 20);
    }
  };
  auto load_tile = [&](int mt0, int buf) {
#pragma unroll
    for (int i = 0; i < NPY; ++i) {
      const int m = mt0 + lry + RPY * i;
      ry[i] = buf_load16(rsY, (m < mend && ook) ? (m * p.Co + o0 + lcy * 8) * 2 : OOB_OFF);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = lrx + 32 * i, m = mt0 + rr;
      const int oy = rowinfo[buf * BKM + rr];
      const bool ok = cok && (unsigned)(oy + tdy) < (unsigned)p.H;
      rx[i] = buf_load16(rsX, ok ? ((m + tdy * p.W) * p.Ci + ci0 + lcx * 8) * 2 : OOB_OFF);
    }
    if (p.halo && t < 16) {     // W > 64: the tile is a 64-pixel piece of one image row; fetch its two neighbours
      const int side = t >> 3, ox0 = mt0 & (p.W - 1);
      const int oy = rowinfo[buf * BKM];
      const bool ok = (side ? (ox0 + 64 < p.W) : (ox0 > 0)) && (unsigned)(oy + tdy) < (unsigned)p.H && (ci0 + (t & 7) * 8) < p.Ci;
      const int m = mt0 + (side ? 64 : -1);
      rh = buf_load16(rsX, ok ? ((m + tdy * p.W) * p.Ci + ci0 + (t & 7) * 8) * 2 : OOB_OFF);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NPY; ++i) *reinterpret_cast<uint4*>(ys + swz256(lry + RPY * i, lcy)) = ry[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = lrx + 32 * i;
      *reinterpret_cast<uint4*>(xs + swzx(r + 2 + 4 * (r >> p.lw), lcx)) = rx[i];
    }
    if (p.halo && t < 16) *reinterpret_cast<uint4*>(xs + swzx((t >> 3) ? 66 : 1, t & 7)) = rh;
  };

  f32x16_t acc[3][MT];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;

  const int r31 = lane & 31, hi = lane >> 5;
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  int trA[MT][2], trB[3][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 8 * (tg >> 1) + tq + 4 * u;                 // reduction row inside a 16-row k-step
#pragma unroll
    for (int i = 0; i < MT; ++i) trA[i][u] = swz256(row, (wm0 + i * 32) / 8 + 2 * (tg & 1) + (tp >> 1)) + 8 * (tp & 1);
    const int R0 = row + 2 + 4 * (row >> p.lw);
#pragma unroll
    for (int a = 0; a < 3; ++a) trB[a][u] = swzx(R0 + a - 1, wn0 / 8 + 2 * (tg & 1) + (tp >> 1)) + 8 * (tp & 1);
  }
  decode_rows(mbeg, 0);
  __syncthreads();
  if (mbeg < mend) load_tile(mbeg, 0);
  int buf = 0;
  for (int mt0 = mbeg; mt0 < mend; mt0 += BKM, buf ^= 1) {
    __syncthreads();
    store_tile();
    decode_rows(mt0 + BKM, buf ^ 1);
    __syncthreads();
    if (mt0 + BKM < mend) load_tile(mt0 + BKM, buf ^ 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      typedef __attribute__((address_space(3))) bf16x4_t* lds4;
      const int eoff = (16 * s + 4 * ((16 * s) >> p.lw)) * 128;       // image-row offset of k-step s (uniform)
      bf16x8_t a[MT], b[3];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        bf16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(ys + trA[i][0] + s * 4096));
        bf16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(ys + trA[i][1] + s * 4096));
        a[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(xs + trB[k][0] + eoff));
        bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(xs + trB[k][1] + eoff));
        b[k] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[k][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[k], acc[k][i], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int c = ci0 + wn0 + r31;
        if (o < p.Co && c < p.Ci) out[(size_t)o * p.ldw + (kh * 3 + k) * p.Ci + c] = acc[k][i][r];
      }
}
template <int MT>
__global__ __launch_bounds__(256, 3) void wgrad_kw_kernel(const WgradKwArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[WgradKwSmem<MT>::kBytes];
  wgrad_kw_body<MT>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}
// The 3x3 / stride-1 weight gradients of one ResNet stage in ONE launch (bf16).  Alone, such a layer offers 24 .. 48 output tiles:
// it splits its pixel range 16 .. 256 ways to reach a few hundred blocks, which then sit unevenly on the 256 CUs (384 blocks:
// half the CUs carry two), run 8 .. 16 K steps each (mostly prologue and fp32 slab stores) and leave 16 .. 256 slabs to a reduce
// launch of their own.  Grouped, the tiles of all the stage's layers fill the chip together with a few long splits per layer and
// one grouped slab reduction.  Argument blocks by value in the kernel-argument segment (graph-capturable), as wgrad_group_kernel.
#define WGK_MAX 16
struct WgradKwGroupArgs { WgradKwArgs p[WGK_MAX]; int bstart[WGK_MAX + 1]; int n; };
template <int MT>
__global__ __launch_bounds__(256, 3) void wgrad_kw_group_kernel(const WgradKwGroupArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[WgradKwSmem<MT>::kBytes];
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.bstart[i + 1]) ++i;         // block-uniform scan of <= 16 entries
  wgrad_kw_body<MT>(g.p[i], (int)blockIdx.x - g.bstart[i], g.bstart[i + 1] - g.bstart[i], smem);
}

// ------------------------------------------------------------------------------------ wgrad, 3x3 / 4x4 stride 2 pad 1 (bf16)
// Same idea for stride 2 (the strided convs of the backbone / adversarial heads and the conv-form of the 4x4 transposed
// convs): tap kw of output pixel ox reads input column 2*ox + kw - 1, so the taps fall on TWO column-parity images of the
// input row -- E[j] = x[2j], O[j] = x[2j+1] -- as shifted reads: kw0 = O[ox-1], kw1 = E[ox], kw2 = O[ox], kw3 = E[ox+1].
// Per 64-pixel step a block stages one DY tile and the two parity tiles (as much X data as ONE tap of the generic kernel)
// and multiplies 3 (4) taps out of them.
struct WgradKw2Args {
  const void* X; const void* DY; float* out;
  int H, W, Ho, Ci, Co;
  int lw, lwo;                  // log2(min(Wo,64)), log2(Wo)
  int M, rows_per_split, ldw;
  long slab_stride;
  int nto, nci;
  unsigned x_bytes, dy_bytes;
  FastDiv dHo;
};

template <int MT, int KW>
__global__ __launch_bounds__(256, KW == 4 ? 2 : 3) void wgrad_kw2_kernel(const WgradKw2Args p) {
  constexpr int BO = 64 * MT, BKM = 64;
  constexpr int CPRY = BO / 8, RPY = 256 / CPRY, NPY = BKM / RPY;
  constexpr int XROWS = 64 + 4 * 8 + 4;
  __shared__ __attribute__((aligned(16))) char smem[BKM * 256 + 2 * XROWS * 128 + 2 * BKM * 8];
  char* ys = smem;
  char* xe = smem + BKM * 256;                 // even input columns
  char* xo = xe + XROWS * 128;                 // odd input columns
  int2* rowinfo = reinterpret_cast<int2*>(smem + BKM * 256 + 2 * XROWS * 128);   // [2][BKM]: {input pixel of (2oy, 2ox), oy}

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ntile = p.nto * KW * p.nci;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / ntile;
  int tile = lid - split * ntile;
  const int ot = tile / (KW * p.nci); tile -= ot * KW * p.nci;
  const int kh = tile / p.nci, cit = tile - kh * p.nci;
  const int o0 = ot * BO, ci0 = cit * 64, tdy = kh - 1;
  const int wm0 = (wave >> 1) * (32 * MT), wn0 = (wave & 1) * 32;

  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X, p.x_bytes), rsY = make_rsrc(p.DY, p.dy_bytes);
  const int lcy = t % CPRY, lry = t / CPRY;
  const int lcx = t & 7, lrx = t >> 3;
  const bool ook = (o0 + lcy * 8) < p.Co;
  const bool cok = (ci0 + lcx * 8) < p.Ci;
  const int mbeg = split * p.rows_per_split;
  const int mend = min(p.M, mbeg + p.rows_per_split);

  for (int i = t; i < 2 * XROWS * 8; i += 256) reinterpret_cast<uint4*>(xe)[i] = make_uint4(0, 0, 0, 0);

  uint4 ry[NPY], re[2], ro[2];
  auto decode_rows = [&](int mt0, int buf) {
    if (t < BKM) {
      const unsigned m = (unsigned)(mt0 + t);
      const unsigned q = m >> p.lwo, ox = m & ((1u << p.lwo) - 1);
      const unsigned n = fd_div(q, p.dHo);
      const int oy = (int)(q - n * p.Ho);
      rowinfo[buf * BKM + t] = (m < (unsigned)p.M) ? make_int2(((int)n * p.H + 2 * oy) * p.W + 2 * (int)ox, oy) : make_int2(0, -(1 << 20));
    }
  };
  auto load_tile = [&](int mt0, int buf) {
#pragma unroll
    for (int i = 0; i < NPY; ++i) {
      const int m = mt0 + lry + RPY * i;
      ry[i] = buf_load16(rsY, (m < mend && ook) ? (m * p.Co + o0 + lcy * 8) * 2 : OOB_OFF);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int2 ri = rowinfo[buf * BKM + lrx + 32 * i];
      const bool ok = cok && (unsigned)(2 * ri.y + tdy) < (unsigned)p.H;
      const int off = ((ri.x + tdy * p.W) * p.Ci + ci0 + lcx * 8) * 2;
      re[i] = buf_load16(rsX, ok ? off : OOB_OFF);
      ro[i] = buf_load16(rsX, ok ? off + p.Ci * 2 : OOB_OFF);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NPY; ++i) *reinterpret_cast<uint4*>(ys + swz256(lry + RPY * i, lcy)) = ry[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = lrx + 32 * i;
      const int a = swzx(r + 2 + 4 * (r >> p.lw), lcx);
      *reinterpret_cast<uint4*>(xe + a) = re[i];
      *reinterpret_cast<uint4*>(xo + a) = ro[i];
    }
  };

  f32x16_t acc[KW][MT];
#pragma unroll
  for (int a = 0; a < KW; ++a)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;

  const int r31 = lane & 31, hi = lane >> 5;
  const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  int trA[MT][2], trB[KW][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 8 * (tg >> 1) + tq + 4 * u;
#pragma unroll
    for (int i = 0; i < MT; ++i) trA[i][u] = swz256(row, (wm0 + i * 32) / 8 + 2 * (tg & 1) + (tp >> 1)) + 8 * (tp & 1);
    const int R0 = row + 2 + 4 * (row >> p.lw);
    const int ch = wn0 / 8 + 2 * (tg & 1) + (tp >> 1);
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      // kw0 = O[ox-1], kw1 = E[ox], kw2 = O[ox], kw3 = E[ox+1]   (offsets relative to the even image)
      const int shift = k == 0 ? -1 : (k == 3 ? 1 : 0);
      trB[k][u] = swzx(R0 + shift, ch) + 8 * (tp & 1) + ((k & 1) ? 0 : XROWS * 128);
    }
  }
  decode_rows(mbeg, 0);
  __syncthreads();
  if (mbeg < mend) load_tile(mbeg, 0);
  int buf = 0;
  for (int mt0 = mbeg; mt0 < mend; mt0 += BKM, buf ^= 1) {
    __syncthreads();
    store_tile();
    decode_rows(mt0 + BKM, buf ^ 1);
    __syncthreads();
    if (mt0 + BKM < mend) load_tile(mt0 + BKM, buf ^ 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      typedef __attribute__((address_space(3))) bf16x4_t* lds4;
      const int eoff = (16 * s + 4 * ((16 * s) >> p.lw)) * 128;
      bf16x8_t a[MT], b[KW];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        bf16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(ys + trA[i][0] + s * 4096));
        bf16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(ys + trA[i][1] + s * 4096));
        a[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(xe + trB[k][0] + eoff));
        bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4)(xe + trB[k][1] + eoff));
        b[k] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int k = 0; k < KW; ++k)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[k][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[k], acc[k][i], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  float* out = p.out + (size_t)split * p.slab_stride;
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int c = ci0 + wn0 + r31;
        if (o < p.Co && c < p.Ci) out[(size_t)o * p.ldw + (kh * KW + k) * p.Ci + c] = acc[k][i][r];
      }
}

// Fixed-order slab reduction.  SL "slab lanes" share each float4 column: lane j sums slabs j, j+SL, ... (4 loads in
// flight), the SL partial sums are then added in lane order -- the order depends only on (S, SL), never on timing.
template <int SL>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, long n, int S,
                                                           long stride, int accumulate) {
  constexpr int COLS = 256 / SL;
  __shared__ float4 part[SL > 1 ? 256 : 1];
  const long n4 = n >> 2;
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  for (long i0 = (long)blockIdx.x * COLS; i0 < n4; i0 += (long)gridDim.x * COLS) {
    const long i = i0 + col;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      int s = sl;
      for (; s + 3 * SL < S; s += 4 * SL) {
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = reinterpret_cast<const float4*>(slabs + (size_t)(s + u * SL) * stride)[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) { v.x += a[u].x; v.y += a[u].y; v.z += a[u].z; v.w += a[u].w; }
      }
      for (; s < S; s += SL) { const float4 a = reinterpret_cast<const float4*>(slabs + (size_t)s * stride)[i]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    }
    if constexpr (SL > 1) {
      __syncthreads();
      part[threadIdx.x] = v;
      __syncthreads();
      if (sl == 0 && i < n4) {
        float4 r = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < SL; ++j) { const float4 q = part[j * COLS + col]; r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w; }
        reinterpret_cast<float4*>(out)[i] = r;
      }
    } else {
      if (i < n4) {
        if (accumulate) { const float4 o = reinterpret_cast<const float4*>(out)[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        reinterpret_cast<float4*>(out)[i] = v;
      }
    }
  }
}
void launch_slab_reduce(const float* ws, float* dw, long n, int S, long stride, int accumulate, hipStream_t st) {
  const long n4 = n / 4;
  // enough slab lanes that ~64K threads stream, but never more lanes than slabs
  int SL = 1;
  while (SL < 16 && n4 * SL < 65536 && SL * 2 <= S) SL *= 2;
  const int cols = 256 / SL;
  dim3 grid(cdiv(n4, cols));
  char lab[64];
  if (prof_on()) snprintf(lab, sizeof(lab), "slab_reduce n%ld S%d", n, S);
  ProfScope ps(st, 0.0, 4.0 * n * (S + 1), 2, prof_on() ? lab : nullptr);
  if (SL == 1) hipLaunchKernelGGL(slab_reduce_kernel<1>, grid, dim3(256), 0, st, ws, dw, n, S, stride, accumulate);
  else if (SL == 2) hipLaunchKernelGGL(slab_reduce_kernel<2>, grid, dim3(256), 0, st, ws, dw, n, S, stride, accumulate);
  else if (SL == 4) hipLaunchKernelGGL(slab_reduce_kernel<4>, grid, dim3(256), 0, st, ws, dw, n, S, stride, accumulate);
  else if (SL == 8) hipLaunchKernelGGL(slab_reduce_kernel<8>, grid, dim3(256), 0, st, ws, dw, n, S, stride, accumulate);
  else hipLaunchKernelGGL(slab_reduce_kernel<16>, grid, dim3(256), 0, st, ws, dw, n, S, stride, accumulate);
}

// plain zero-fill (own kernel rather than hipMemsetAsync: a kernel node replays identically inside HIP graphs)
__global__ void zero_fill_kernel(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0, 0, 0, 0);
}

// ------------------------------------------------------------------------------------ host side
static int ilog2_exact(int v) { int s = 0; while ((1 << s) < v) ++s; return ((1 << s) == v) ? s : -1; }

template <typename T, int BM, int BN, bool SMALL_C, int WGM, int WGN, bool HM_OUT, bool DMA, int EPI, bool KW3 = false, bool CAT = false, int KG = 1>
static void launch_gather_epi(const GatherArgs& a, hipStream_t st) {
  using SM = GatherSmem<T, BM, BN, DMA ? 2 : 1, KW3>;
  constexpr int smem = SM::kBytes + (KG - 1) * 2 * SM::kStage;
  auto kern = gather_gemm_kernel<T, BM, BN, SMALL_C, WGM, WGN, HM_OUT, DMA, EPI, KW3, CAT, KG>;
  static bool attr_set = false;   // raise the dynamic-LDS cap once per instantiation
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(64 * WGM * WGN * KG), smem, st, a);
}

template <typename T, int BM, int BN, bool SMALL_C, int WGM = 2, int WGN = 2, bool HM_OUT = false, bool DMA = false, bool KW3 = false, bool CAT = false, int KG = 1>
static void launch_gather(GatherArgs& a, hipStream_t st) {
  a.ntn = cdiv(a.Nout, BN);
  int mx = 0;
  for (int i = 0; i < a.nphase; ++i) { a.ph[i].ntm = cdiv(a.ph[i].M, BM); if (a.ph[i].ntm > mx) mx = a.ph[i].ntm; }
  a.ntiles = a.nphase * mx * a.ntn;
  a.stat_slices = 0;
  if (a.stat_partial) {
    bool even = !HM_OUT && !a.residual && !a.accumulate;        // phases of unequal tile count would leave unwritten slices
    for (int i = 0; i < a.nphase; ++i) even = even && a.ph[i].ntm == mx;
    if (even && (size_t)a.nphase * mx * a.Nout * 3 * sizeof(float) <= a.stat_bytes) a.stat_slices = a.nphase * mx;
    else a.stat_partial = nullptr;
  }
  if (a.bnb_partial) {
    bool even = !HM_OUT && !a.stat_partial;
    for (int i = 0; i < a.nphase; ++i) even = even && a.ph[i].ntm == mx;
    if (even && (size_t)a.nphase * mx * a.Nout * 2 * sizeof(float) <= a.stat_bytes) a.stat_slices = a.nphase * mx;
    else a.bnb_partial = nullptr;
  }
  constexpr bool EXTRAS = !HM_OUT && BM <= 128 && !CAT && KG == 1;     // the BatchNorm-backward epilogue exists for the regular tiles only
  constexpr bool STATS = !HM_OUT && (BM <= 128 || (BM == 256 && BN == 128) || (BM == 256 && BN == 256 && DMA));   // statistics: also the 256x128 macro tile and the 256x256 LDS-DMA build
  if (!EXTRAS && a.bnb_partial) a.bnb_partial = nullptr;
  static const bool stats256 = !(getenv("MI355_STATS_256") && atoi(getenv("MI355_STATS_256")) == 0);      // A/B switch
  if ((!STATS || (BM == 256 && !stats256)) && a.stat_partial) { a.stat_partial = nullptr; a.stat_slices = 0; }
  if (!a.stat_partial && !a.bnb_partial) a.stat_slices = 0;
  if constexpr (EXTRAS) {
    if (a.bnb_partial) { launch_gather_epi<T, BM, BN, SMALL_C, WGM, WGN, HM_OUT, DMA, 2>(a, st); return; }     // (EPI 2 has no KW3 build)
  }
  if constexpr (STATS) {
    if (a.stat_partial) { launch_gather_epi<T, BM, BN, SMALL_C, WGM, WGN, HM_OUT, DMA, 1, KW3, CAT, KG>(a, st); return; }
  }
  launch_gather_epi<T, BM, BN, SMALL_C, WGM, WGN, HM_OUT, DMA, 0, KW3, CAT, KG>(a, st);
}

// Do the 256 x 256 tiles of this launch fill whole rounds of 256 CUs (one 8-wave block per CU)?  mode 1: exactly one round of
// >= tmin tiles; mode 2 (experiment): up to four rounds, the last one with >= tmin tiles, phases of a strided launch counted separately.
static bool t256_fits(const GatherArgs& a, long Mtot, int mode, long tmin) {
  (void)Mtot;
  long t = 0;
  for (int i = 0; i < a.nphase; ++i) t += cdiv(a.ph[i].M, 256L) * (a.Nout / 256);
  if (mode < 2) return t >= tmin && t <= 256;
  const long rem = t % 256;
  return t >= tmin && t <= 1024 && (rem == 0 || rem >= tmin);
}
template <typename T>
static int dispatch_gather(GatherArgs& a, hipStream_t st) {
  constexpr int CH = MmaTraits<T>::CH;
  if (a.Ci % CH) MI_FAIL(MI355_EINVAL, "gather: Ci=%d not a multiple of %d", a.Ci, CH);
  a.cshift = ilog2_exact(a.Ci / CH);
  if (a.cshift < 0) MI_FAIL(MI355_EINVAL, "gather: Ci/%d must be a power of two (Ci=%d)", CH, a.Ci);
  if (a.Nout % CH) MI_FAIL(MI355_EINVAL, "gather: Nout=%d not a multiple of %d", a.Nout, CH);
  if (a.nphase < 1 || a.nphase > 4) MI_FAIL(MI355_EINVAL, "gather: nphase=%d", a.nphase);
  long Mtot = 0, ntaps_tot = 0; int kchunks = 0; double flops = 0.0;
  for (int i = 0; i < a.nphase; ++i) {
    Mtot += a.ph[i].M; ntaps_tot += a.ph[i].ntaps;
    if ((a.ph[i].ntaps << a.cshift) > kchunks) kchunks = a.ph[i].ntaps << a.cshift;
    flops += 2.0 * a.ph[i].M * (double)a.Nout * a.ph[i].ntaps * a.Ci;
  }
  {
    // A spans [images][Hi][Wi][Ci]; images = M / (OHp*OWp)
    const long imgs = a.ph[0].M / ((long)a.ph[0].OHp * a.ph[0].OWp);
    a.a_bytes = (unsigned)(imgs * a.Hi * a.Wi * a.Ci * (long)sizeof(T));
    a.b_bytes = (unsigned)((long)a.Nout * a.ldb * (long)sizeof(T));
  }
  const bool small = (a.Ci / CH) < 8;
  const long kavg = (ntaps_tot << a.cshift) / a.nphase;   // phases of a strided dgrad differ in length
  (void)kchunks;
  // algorithmic bytes: every input element, weight and output element once
  if (a.A2) flops += 2.0 * Mtot * (double)a.Nout * a.c2;
  ProfScope ps(st, flops, (double)a.a_bytes + (double)a.b_bytes * ntaps_tot / (a.ldb / a.Ci) + (double)Mtot * a.Nout * sizeof(T) +
                              (a.A2 ? (double)a.a2_bytes + a.b2_bytes : 0.0));
  static const int force = getenv("MI355_TILE") ? atoi(getenv("MI355_TILE")) : -1;   // experiment switch
  static const int dma_mode = getenv("MI355_DMA") ? atoi(getenv("MI355_DMA")) : 1;
  if (a.A2) {
    // concatenated-K forward: the tile choices of the plain path that matter for the two layers that use it (1x1 and 3x3 / stride 2,
    // 256 -> 256 channels), register-staged or LDS-DMA ring
    if (small || a.nphase != 1 || a.out_sx != 1 || a.out_sy != 1 || a.bnb_partial || a.residual || a.accumulate)
      MI_FAIL(MI355_EINVAL, "concat-K forward: plain single-phase forward conv with >= 8 input chunks only");
    if (a.c2 < CH || a.c2 % CH || a.c2 > MmaTraits<T>::BK) MI_FAIL(MI355_EINVAL, "concat-K forward: c2=%d must be a multiple of %d and <= %d", a.c2, CH, MmaTraits<T>::BK);
    const long t128 = cdiv(Mtot, 128L) * cdiv(a.Nout, 128);
    static const int cat_tile = getenv("MI355_CAT_TILE") ? atoi(getenv("MI355_CAT_TILE")) : 0;     // experiment switch
    static const int t256d_cat = getenv("MI355_T256D") ? atoi(getenv("MI355_T256D")) : 2;
    static const long t256d_cmin = getenv("MI355_T256D_MIN") ? atol(getenv("MI355_T256D_MIN")) : 192;
    if (t256d_cat >= 2 && sizeof(T) == 2 && a.Nout % 256 == 0 && dma_mode == 1 && t256_fits(a, Mtot, t256d_cat, t256d_cmin)) {
      if constexpr (sizeof(T) == 2) launch_gather<T, 256, 256, false, 2, 4, false, true, false, true>(a, st);
    }
    else if (cat_tile == 1 && a.Nout > 64) launch_gather<T, 128, 128, false, 2, 2, false, false, false, true>(a, st);
    else if (cat_tile == 2 && a.Nout > 64) launch_gather<T, 128, 128, false, 2, 2, false, true, false, true>(a, st);
    else if (cat_tile == 3 && a.Nout > 64) launch_gather<T, 64, 128, false, 2, 2, false, false, false, true>(a, st);
    else if (a.Nout <= 64) launch_gather<T, 64, 64, false, 2, 2, false, false, false, true>(a, st);
    else if ((dma_mode == 2) || (dma_mode == 1 && ((t128 >= 512 && kavg >= 128) || (t128 >= 256 && kavg >= 256)))) launch_gather<T, 128, 128, false, 2, 2, false, true, false, true>(a, st);
    else if (t128 >= 512 && kavg <= 32 && sizeof(T) == 2) launch_gather<T, 64, 128, false, 2, 2, false, false, false, true>(a, st);
    else if (t128 >= 512) launch_gather<T, 128, 128, false, 2, 2, false, false, false, true>(a, st);
    else if (cdiv(Mtot, 64L) * cdiv(a.Nout, 128) >= 512) launch_gather<T, 64, 128, false, 2, 2, false, false, false, true>(a, st);
    else launch_gather<T, 64, 64, false, 2, 2, false, false, false, true>(a, st);
    MI_CHECK_LAUNCH("gather_gemm_cat");
    return MI355_OK;
  }
  if constexpr (sizeof(T) == 2) {
    if (pgemm_eligible(a, 2)) return dispatch_pgemm(a, st);       // 1x1 / unit stride: the persistent pipelined GEMM (pgemm.hip)
  }
  if (small) { launch_gather<T, 128, 64, true>(a, st); }
  else if (force == 0) launch_gather<T, 128, 128, false>(a, st);
  else if (force == 1) launch_gather<T, 64, 128, false>(a, st);
  else if (force == 2) launch_gather<T, 128, 64, false>(a, st);
  else if (force == 3) launch_gather<T, 64, 64, false>(a, st);
  else {
    const long t128 = cdiv(Mtot, 128L) * cdiv(a.Nout, 128);
    // 3x3 / unit stride / same-size maps of a power-of-two width <= 128: the A-tile-sharing variant (see KW3 above)
    static const int kw3_on = getenv("MI355_KW3") ? atoi(getenv("MI355_KW3")) : 1;
    // A/B switch: 256 x 256 LDS-DMA tiles.  1: launches of ONE round of tiles (30.41 / 30.42 -> 30.15 / 30.18 ms; with 128 .. 191 tiles too:
    // slower); 2 (default): also up to four full rounds, the phases of strided input gradients / transposed convs, the concat-K forward
    static const int t256d = getenv("MI355_T256D") ? atoi(getenv("MI355_T256D")) : 2;
    static const long t256d_min = getenv("MI355_T256D_MIN") ? atol(getenv("MI355_T256D_MIN")) : 192;
    static const long t256d_kmin = getenv("MI355_T256D_KMIN") ? atol(getenv("MI355_T256D_KMIN")) : 32;
    static const int splitk_on = getenv("MI355_SPLITK") ? atoi(getenv("MI355_SPLITK")) : 2;                 // A/B switch (1: 128 x 128 tiles only)
    const long t64 = cdiv(Mtot, 64L) * cdiv(a.Nout, 128);
    static const long splitk_min = getenv("MI355_SPLITK_MIN") ? atol(getenv("MI355_SPLITK_MIN")) : 128;
    static const long splitk_max = getenv("MI355_SPLITK_MAX") ? atol(getenv("MI355_SPLITK_MAX")) : 320;
    static const long splitk_kmin = getenv("MI355_SPLITK_KMIN") ? atol(getenv("MI355_SPLITK_KMIN")) : 128;      // shortest K taken, in 16-byte chunks (128: the 1x1 convs with K = 1024 at 16x16 too, 30.88 / 30.81 -> 30.76 / 30.76 ms)
    static const int kw3_n64 = getenv("MI355_KW3_N64") ? atoi(getenv("MI355_KW3_N64")) : 1;      // A/B switch: the variant for 64 output channels
    bool kw3 = kw3_on && sizeof(T) == 2 && a.nphase == 1 && a.ph[0].ntaps == 9 && a.in_sx == 1 && a.in_sy == 1 && a.out_sx == 1 &&
               a.out_sy == 1 && a.ph[0].OWp == a.Wi && a.ph[0].OHp == a.Hi && a.Wo == a.Wi && a.Ho == a.Hi && a.Wi >= 8 &&
               a.Wi <= 128 && ilog2_exact(a.Wi) >= 0 && (a.Nout > 64 || (kw3_n64 && a.Nout == 64)) && a.Ci % 64 == 0 && !a.bnb_partial;
    for (int g = 0; g < 3 && kw3; ++g) {
      const Tap* tp = a.taps + a.ph[0].tap0 + 3 * g;
      int seen = 0;
      for (int k = 0; k < 3; ++k) { if (tp[k].dy != tp[0].dy || tp[k].dx < -1 || tp[k].dx > 1) kw3 = false; else seen |= 1 << (tp[k].dx + 1); }
      if (seen != 7 || tp[0].dy < -1 || tp[0].dy > 1) kw3 = false;
    }
    if (kw3) a.lw = ilog2_exact(a.Wi);
    if (a.Nout <= 64) {
      // 3x3 64 -> 64 on the large maps (layer1 of the ResNets: 2048 tiles of 128 x 64): the shared-A-tile variant here too -- two
      // thirds of what a 128 x 64 tile stages per tap is the A tile
      if (kw3 && (cdiv(Mtot, 128L) >= 2048 || kw3_on == 2)) { if constexpr (sizeof(T) == 2) launch_gather<T, 128, 64, false, 2, 2, false, false, true>(a, st); }
      else if (cdiv(Mtot, 128L) >= 512) launch_gather<T, 128, 64, false>(a, st); else launch_gather<T, 64, 64, false>(a, st);
    } else if (sizeof(T) == 2 && getenv("MI355_T256") && a.Nout % 256 == 0 && cdiv(Mtot, 256L) * (a.Nout / 256) >= 256) launch_gather<T, 256, 256, false, 2, 4>(a, st);
    // one 256 x 256 tile per CU on the LDS-DMA ring (8 waves, 128 accumulators each): half the bytes through L1 per MFMA of the 128 x 128
    // tiles, for launches that offer one round of such tiles
    // (not the accumulating epilogues: their read-modify-write of a 128-KB tile has no second block on the CU to hide behind --
    //  1x1 1024 -> 256 @16x16 input gradient + masked accumulate 20.8 -> 25.2 us, 256 -> 256 @64x64 + accumulate 93 -> 112 us)
    else if (t256d && sizeof(T) == 2 && (a.nphase == 1 || t256d >= 2) && a.Nout % 256 == 0 && dma_mode == 1 && !a.bnb_partial && !a.accumulate &&
             t256_fits(a, Mtot, t256d, t256d_min) && kavg >= t256d_kmin && !(kw3 && t128 >= 2048)) {      // (the big 3x3 layers keep the shared-A-tile kernels: 309.7 vs 312.7 us)
      if constexpr (sizeof(T) == 2) launch_gather<T, 256, 256, false, 2, 4, false, true>(a, st);
    }
    // LDS-DMA ring for K-heavy layers (>= 16 K-tiles): +9..12 % on the 3x3 / 4x4 convs, but -15 % on short-K 1x1 convs
    // (2 blocks/CU instead of 3), so those keep the register-staged form.  MI355_DMA=0 disables, =2 forces (tests).
    // (measured: 334 -> 310 us forward, 324 -> 312 us dgrad on 256->256 @64x64; at 1024 tiles the LDS-DMA ring still wins)
    // 256x128 macro tile (128 accumulators per wave, 2 blocks/CU, 0.21 KB of L1 traffic per MFMA): 313 -> 302 us on the 64x64 layers
    else if (kw3 && t128 >= 4096 && kw3_on != 2 && kw3_on != 4) { if constexpr (sizeof(T) == 2) launch_gather<T, 256, 128, false, 2, 2, false, false, true>(a, st); }
    else if (kw3 && (t128 >= 2048 || kw3_on == 2)) { if constexpr (sizeof(T) == 2) launch_gather<T, 128, 128, false, 2, 2, false, false, true>(a, st); }
    // (also the K-heavy mid-size layers, 256 .. 511 tiles with K >= 2048: 3x3 256->256 @16x16 36.2 -> 32.7 us; a 3-stage ring with
    //  two tiles in flight and counted vmcnt measured 33.5 us there: the per-CU fill rate, not latency, bounds these layers)
    // one 128 x 128 tile per CU or fewer and a long K: two K groups per workgroup (KG above) -- 3x3 256 -> 256 @16x16 and kin
    // fewer 128 x 128 tiles than CUs (the 8x8 maps: 128 of them would leave half the chip idle): 64-row tiles, two K groups each
    else if (splitk_on >= 2 && dma_mode == 1 && t128 < 256 && t64 >= splitk_min && t64 <= 384 && kavg >= splitk_kmin && !a.bnb_partial) launch_gather<T, 64, 128, false, 2, 2, false, true, false, false, 2>(a, st);
    else if (splitk_on && dma_mode == 1 && t128 >= splitk_min && t128 <= splitk_max && kavg >= splitk_kmin && !a.bnb_partial) launch_gather<T, 128, 128, false, 2, 2, false, true, false, false, 2>(a, st);
    else if ((dma_mode == 2 && a.Nout > 64) || (dma_mode == 1 && ((t128 >= 512 && kavg >= 128) || (t128 >= 256 && kavg >= 256)))) launch_gather<T, 128, 128, false, 2, 2, false, true>(a, st);
    // short-K 1x1 convs at large M are all prologue / epilogue and HBM-bound: more, smaller blocks in flight win
    // (64->256 @64x64: 54.5 -> 47.0 us, 256->256: 79.6 -> 68.9 us)
    else if (t128 >= 512 && kavg <= 32 && sizeof(T) == 2) launch_gather<T, 64, 128, false>(a, st);
    else if (t128 >= 512) launch_gather<T, 128, 128, false>(a, st);   // (128x256 tile with 8 waves measured slower: 687 vs 755 TFLOP/s)
    else if (cdiv(Mtot, 64L) * cdiv(a.Nout, 128) >= 512) launch_gather<T, 64, 128, false>(a, st);
    else launch_gather<T, 64, 64, false>(a, st);
  }
  MI_CHECK_LAUNCH("gather_gemm");
  return MI355_OK;
}

// crop_ok: the entry point also takes a CROPPED output -- the top-left Ho x Wo outputs of a unit-stride conv (forward and weight
// gradient only).  That is how the 7x7 / stride-2 stem runs as a 4x4 conv over the space-to-depth image (window rows -2 .. +1:
// pad 2, the 129th output row / column never computed).
static int check_desc(const mi355_conv_desc* d, bool crop_ok = false) {
  if (!d) MI_FAIL(MI355_EINVAL, "null conv desc");
  if (d->dtype != MI355_F32 && d->dtype != MI355_BF16 && d->dtype != MI355_FP8) MI_FAIL(MI355_EINVAL, "bad dtype %d", d->dtype);
  if (d->kh * d->kw > MAX_TAPS || d->kh < 1 || d->kw < 1) MI_FAIL(MI355_EINVAL, "unsupported kernel %dx%d", d->kh, d->kw);
  if (d->stride < 1 || d->stride > 2) MI_FAIL(MI355_EINVAL, "unsupported stride %d", d->stride);
  const int fho = (d->Hi + 2 * d->pad - d->kh) / d->stride + 1, fwo = (d->Wi + 2 * d->pad - d->kw) / d->stride + 1;
  const bool cropped = crop_ok && d->stride == 1 && d->kh > 1 && d->dtype != MI355_FP8 && d->Ho >= 1 && d->Wo >= 1 && d->Ho <= fho && d->Wo <= fwo;
  if (!cropped && (d->Ho != fho || d->Wo != fwo))
    MI_FAIL(MI355_EINVAL, "conv desc: output size %dx%d inconsistent with input %dx%d k%d s%d p%d", d->Ho, d->Wo, d->Hi, d->Wi, d->kh, d->stride, d->pad);
  // the kernels form signed 32-bit BYTE offsets and 32-bit buffer-resource sizes: bound bytes, not elements
  const long esz = d->dtype == MI355_F32 ? 4 : 2;      // (fp8 operands, bf16 results: bound by the wider side)
  if ((long)d->N * d->Hi * d->Wi * d->Ci * esz >= (1L << 31) || (long)d->N * d->Ho * d->Wo * d->Co * esz >= (1L << 31))
    MI_FAIL(MI355_EINVAL, "tensor too large for 32-bit byte offsets (%ld / %ld bytes): split the batch",
            (long)d->N * d->Hi * d->Wi * d->Ci * esz, (long)d->N * d->Ho * d->Wo * d->Co * esz);
  return MI355_OK;
}

static void set_bnb(GatherArgs& a, const mi355_bn_bwd_src* bn, float* partial, size_t partial_bytes) {
  a.bnb_x = bn->x; a.bnb_y = bn->relu ? bn->y : nullptr;
  a.bnb_mean = bn->save_mean; a.bnb_invstd = bn->save_invstd; a.bnb_gamma = bn->gamma; a.bnb_beta = bn->beta;
  a.bnb_relu = bn->relu ? (bn->y ? 1 : 2) : 0;
  a.bnb_partial = partial; a.stat_bytes = partial_bytes;
}
static int check_bnb(const mi355_bn_bwd_src* bn, const float* partial, const int* nslices) {
  if (!bn || !partial || !nslices || !bn->x || !bn->save_mean || !bn->save_invstd)
    MI_FAIL(MI355_EINVAL, "bnbwd fusion: bn source / partial / nslices must be given");
  if (bn->relu && !bn->y && (!bn->gamma || !bn->beta)) MI_FAIL(MI355_EINVAL, "bnbwd fusion: relu without y needs gamma and beta");
  return MI355_OK;
}
// fp8 operand launches: device scalars undoing the operand scales, and the format of the gathered operand
struct Fp8Extra { const float* descale_a; const float* descale_b; int a_fmt; };
struct CatExtra { const void* x2; const void* w2; const float* bias2; int c2; };
static int conv_fwd_impl(const mi355_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y,
                         float* partial, size_t partial_bytes, int* nslices, void* stream, const mi355_bn_bwd_src* bn = nullptr,
                         const Fp8Extra* f8 = nullptr, int relu = 0, const CatExtra* cat = nullptr) {
  if (int e = check_desc(d, true)) return e;
  if ((d->dtype == MI355_FP8) != (f8 != nullptr)) MI_FAIL(MI355_EINVAL, "fp8 descriptors go through the *_fp8 entry points (and only they)");
  if (relu && (bn || f8 || partial)) MI_FAIL(MI355_EINVAL, "conv_fwd: the fused ReLU is an inference epilogue (no statistics / BatchNorm-backward / fp8 variant)");
  GatherArgs a; memset(&a, 0, sizeof(a));
  if (prof_on()) prof_set_tag("fwd%s k%ds%d %d>%d @%dx%d n%d%s%s%s", f8 ? "8" : "", d->kh, d->stride, d->Ci, d->Co, d->Hi, d->Wi, d->N, partial ? " +stats" : "",
                              residual ? " +res" : "", cat ? " +cat" : "");
  a.relu = relu ? 1 : 0;
  a.A = x; a.B = w; a.D = y; a.bias = bias; a.residual = residual; a.scale = nullptr;
  a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.in_sy = a.in_sx = d->stride;
  a.Ho = d->Ho; a.Wo = d->Wo; a.out_sy = a.out_sx = 1;
  a.Nout = d->Co; a.ldd = d->Co; a.ldb = d->kh * d->kw * d->Ci; a.accumulate = 0;
  a.nphase = 1; a.ph[0].OHp = d->Ho; a.ph[0].OWp = d->Wo; a.ph[0].M = d->N * d->Ho * d->Wo; a.ph[0].ntaps = d->kh * d->kw;
  if (bn) set_bnb(a, bn, partial, partial_bytes);
  else { a.stat_partial = partial; a.stat_bytes = partial_bytes; }
  if (cat) {
    const long esz = d->dtype == MI355_F32 ? 4 : 2;
    a.A2 = cat->x2; a.B2 = cat->w2; a.bias2 = cat->bias2; a.c2 = cat->c2;
    a.a2_bytes = (unsigned)((long)d->N * d->Ho * d->Wo * cat->c2 * esz); a.b2_bytes = (unsigned)((long)d->Co * cat->c2 * esz);
  }
  for (int i = 0; i < d->kh; ++i)
    for (int j = 0; j < d->kw; ++j) { Tap& t = a.taps[i * d->kw + j]; t.dy = (int8_t)(i - d->pad); t.dx = (int8_t)(j - d->pad); t.widx = (int16_t)(i * d->kw + j); }
  int e;
  if (f8) { a.scale2 = f8->descale_a; a.scale3 = f8->descale_b; a.a_fmt = f8->a_fmt; e = dispatch_gather_fp8(a, as_stream(stream)); }
  else e = d->dtype == MI355_BF16 ? dispatch_gather<bf16_t>(a, as_stream(stream)) : dispatch_gather<float>(a, as_stream(stream));
  if (nslices) *nslices = a.stat_slices;
  return e;
}
extern "C" int mi355_conv_fwd_fp8(const mi355_conv_desc* d, const void* x8, int x_fmt, const void* w8, const float* descale_x,
                                  const float* descale_w, const float* bias, const void* residual, void* y, float* partial,
                                  size_t partial_bytes, int* nslices, void* stream) {
  if (!descale_x || !descale_w || (x_fmt != 0 && x_fmt != 1)) MI_FAIL(MI355_EINVAL, "conv_fwd_fp8: descale scalars / format missing");
  if (nslices) *nslices = 0;
  Fp8Extra f8{descale_x, descale_w, x_fmt};
  return conv_fwd_impl(d, x8, w8, bias, residual, y, partial, partial_bytes, nslices, stream, nullptr, &f8);
}
extern "C" int mi355_conv_fwd(const mi355_conv_desc* d, const void* x, const void* w, const float* bias,
                              const void* residual, void* y, void* stream) {
  return conv_fwd_impl(d, x, w, bias, residual, y, nullptr, 0, nullptr, stream);
}
// inference: y = act(conv(x) + bias + residual) -- with the BatchNorm that follows folded into w / bias by the caller
extern "C" int mi355_conv_fwd_act(const mi355_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual,
                                  int relu, void* y, void* stream) {
  return conv_fwd_impl(d, x, w, bias, residual, y, nullptr, 0, nullptr, stream, nullptr, nullptr, relu);
}
extern "C" int mi355_conv_fwd_stats(const mi355_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                                    float* partial, size_t partial_bytes, int* nslices, void* stream) {
  if (!partial || !nslices) MI_FAIL(MI355_EINVAL, "conv_fwd_stats: partial / nslices must be given");
  return conv_fwd_impl(d, x, w, bias, nullptr, y, partial, partial_bytes, nslices, stream);
}
// y = conv(x, w) + x2 * w2^T + bias + bias2 as ONE implicit GEMM (K = kh*kw*Ci + c2): x2 [N*Ho*Wo][c2] at the OUTPUT resolution,
// w2 [Co][c2], both in the descriptor's dtype, c2 a multiple of the 16-byte chunk and <= one K tile (64 bf16 / 32 fp32).
// partial / nslices: nullable pair, BatchNorm statistics of y from the epilogue as in mi355_conv_fwd_stats.
extern "C" int mi355_conv_fwd_cat(const mi355_conv_desc* d, const void* x, const void* w, const float* bias, const void* x2,
                                  const void* w2, const float* bias2, int c2, void* y, float* partial, size_t partial_bytes,
                                  int* nslices, void* stream) {
  if (!x2 || !w2 || c2 < 1) MI_FAIL(MI355_EINVAL, "conv_fwd_cat: second operand pair missing");
  if ((partial == nullptr) != (nslices == nullptr)) MI_FAIL(MI355_EINVAL, "conv_fwd_cat: partial and nslices go together");
  if (d && d->dtype == MI355_FP8) MI_FAIL(MI355_EINVAL, "conv_fwd_cat: bf16 / fp32 descriptors only");
  if (nslices) *nslices = 0;
  CatExtra cat{x2, w2, bias2, c2};
  return conv_fwd_impl(d, x, w, bias, nullptr, y, partial, partial_bytes, nslices, stream, nullptr, nullptr, 0, &cat);
}
// ConvTranspose2d input gradient (= conv-form forward) producing the dy of a BatchNorm: its backward reduction in the epilogue
extern "C" int mi355_conv_fwd_bnbwd(const mi355_conv_desc* d, const void* x, const void* w, void* y, const mi355_bn_bwd_src* bn,
                                    float* partial, size_t partial_bytes, int* nslices, void* stream) {
  if (int e = check_bnb(bn, partial, nslices)) return e;
  return conv_fwd_impl(d, x, w, nullptr, nullptr, y, partial, partial_bytes, nslices, stream, bn);
}
// capacity that always suffices for the fused statistics of a conv output / deconv output (smallest tile = 64 rows)
extern "C" size_t mi355_conv_stats_bytes(long rows, int C) { return (size_t)(rows / 64 + 8) * C * 3 * sizeof(float); }

// 1x1 conv C -> K (K <= 32) written as NCHW fp32 heat-maps: y[n][k][p] = bias[k] + sum_c x[n*HW+p][c] * w[k][c]
template <typename T>
static int heatmap_conv(const void* x, const void* w, const float* bias, float* y, int N, int HW, int C, int K, hipStream_t st) {
  constexpr int CH = MmaTraits<T>::CH;
  GatherArgs a; memset(&a, 0, sizeof(a));
  a.A = x; a.B = w; a.D = y; a.bias = bias;
  a.Hi = 1; a.Wi = HW; a.Ci = C; a.in_sy = a.in_sx = 1;
  a.Ho = 1; a.Wo = HW; a.out_sy = a.out_sx = 1; a.ldd = 1; a.hw = HW;
  a.Nout = K; a.ldb = C;
  a.nphase = 1; a.ph[0].OHp = 1; a.ph[0].OWp = HW; a.ph[0].M = N * HW; a.ph[0].ntaps = 1;
  if (C % CH) MI_FAIL(MI355_EINVAL, "conv1x1_heatmap: C=%d not a multiple of %d", C, CH);
  a.cshift = ilog2_exact(C / CH);
  if (a.cshift < 3) MI_FAIL(MI355_EINVAL, "conv1x1_heatmap: C/%d must be a power of two >= 8 (C=%d)", CH, C);
  a.a_bytes = (unsigned)((long)N * HW * C * (long)sizeof(T));
  a.b_bytes = (unsigned)((long)K * C * (long)sizeof(T));
  if (prof_on()) prof_set_tag("hm1x1 %d>%d hw%d n%d", C, K, HW, N);
  ProfScope ps(st, 2.0 * N * HW * (double)K * C, (double)a.a_bytes + a.b_bytes + 4.0 * N * HW * K);
  launch_gather<T, 128, 32, false, 4, 1, true>(a, st);
  MI_CHECK_LAUNCH("conv1x1_heatmap");
  return MI355_OK;
}
extern "C" int mi355_conv1x1_heatmap(const void* x, const void* w, const float* bias, float* y, int N, int HW, int C, int K,
                                     int dtype, void* stream) {
  if (!x || !w || !y || N < 1 || HW < 1 || K < 1 || K > 32) MI_FAIL(MI355_EINVAL, "conv1x1_heatmap: bad args");
  if ((long)N * HW * C >= (1L << 31)) MI_FAIL(MI355_EINVAL, "conv1x1_heatmap: tensor too large");
  return dtype == MI355_BF16 ? heatmap_conv<bf16_t>(x, w, bias, y, N, HW, C, K, as_stream(stream))
       : dtype == MI355_F32 ? heatmap_conv<float>(x, w, bias, y, N, HW, C, K, as_stream(stream))
       : (mi355_set_error("bad dtype"), MI355_EINVAL);
}

// conv-form dgrad: dx[n][iy][ix][ci] = sum_{kh,kw,co} dy[n][(iy+p-kh)/s][(ix+p-kw)/s][co] * w[co][kh][kw][ci]
// decomposed into stride^2 phases (iy%s, ix%s), each a unit-stride gather over its own tap subset.
static int conv_dgrad_impl(const mi355_conv_desc* d, const void* dy, const void* wT, const float* bias, const float* scale_dev,
                           int accumulate, void* dx, float* partial, size_t partial_bytes, int* nslices, void* stream,
                           const mi355_bn_bwd_src* bn = nullptr, const Fp8Extra* f8 = nullptr, const void* acc_mask = nullptr,
                           int relu = 0);
extern "C" int mi355_conv_dgrad_fp8(const mi355_conv_desc* d, const void* dy8, int dy_fmt, const void* wT8, const float* descale_dy,
                                    const float* descale_w, const float* scale_dev, int accumulate, void* dx, float* partial,
                                    size_t partial_bytes, int* nslices, void* stream) {
  if (!descale_dy || !descale_w || (dy_fmt != 0 && dy_fmt != 1)) MI_FAIL(MI355_EINVAL, "conv_dgrad_fp8: descale scalars / format missing");
  Fp8Extra f8{descale_dy, descale_w, dy_fmt};
  return conv_dgrad_impl(d, dy8, wT8, nullptr, scale_dev, accumulate, dx, partial, partial_bytes, nslices, stream, nullptr, &f8);
}
// conv input gradient that is the dy of a BatchNorm: that BatchNorm's backward reduction in the epilogue
extern "C" int mi355_conv_dgrad_bnbwd(const mi355_conv_desc* d, const void* dy, const void* wT, const float* scale_dev, int accumulate,
                                      void* dx, const mi355_bn_bwd_src* bn, float* partial, size_t partial_bytes, int* nslices,
                                      void* stream) {
  if (int e = check_bnb(bn, partial, nslices)) return e;
  return conv_dgrad_impl(d, dy, wT, nullptr, scale_dev, accumulate, dx, partial, partial_bytes, nslices, stream, bn);
}
extern "C" int mi355_conv_dgrad(const mi355_conv_desc* d, const void* dy, const void* wT, const float* bias,
                                const float* scale_dev, int accumulate, void* dx, void* stream) {
  return conv_dgrad_impl(d, dy, wT, bias, scale_dev, accumulate, dx, nullptr, 0, nullptr, stream);
}
// dx <- dgrad + (bit of acc_mask set ? dx : 0): the fork of a residual block whose other branch's gradient is still the
// UNMASKED dy of the block's final ReLU -- BatchNorm's backward then need not write the masked copy (mi355_bn_bwd dresidual)
extern "C" int mi355_conv_dgrad_masked_acc(const mi355_conv_desc* d, const void* dy, const void* wT, const float* scale_dev, void* dx,
                                           const void* acc_mask, void* stream) {
  if (!acc_mask) MI_FAIL(MI355_EINVAL, "conv_dgrad_masked_acc: acc_mask is null");
  return conv_dgrad_impl(d, dy, wT, nullptr, scale_dev, 1, dx, nullptr, 0, nullptr, stream, nullptr, nullptr, acc_mask);
}
// inference ConvTranspose2d forward: dx = act(dgrad(dy) + bias) with the following BatchNorm folded into wT / bias
extern "C" int mi355_conv_dgrad_act(const mi355_conv_desc* d, const void* dy, const void* wT, const float* bias, int relu, void* dx,
                                    void* stream) {
  return conv_dgrad_impl(d, dy, wT, bias, nullptr, 0, dx, nullptr, 0, nullptr, stream, nullptr, nullptr, nullptr, relu);
}
// ConvTranspose2d forward (= conv-form dgrad) with the BatchNorm statistics of its output fused into the epilogue
extern "C" int mi355_conv_dgrad_stats(const mi355_conv_desc* d, const void* dy, const void* wT, void* dx, float* partial,
                                      size_t partial_bytes, int* nslices, void* stream) {
  if (!partial || !nslices) MI_FAIL(MI355_EINVAL, "conv_dgrad_stats: partial / nslices must be given");
  return conv_dgrad_impl(d, dy, wT, nullptr, nullptr, 0, dx, partial, partial_bytes, nslices, stream);
}
static int conv_dgrad_impl(const mi355_conv_desc* d, const void* dy, const void* wT, const float* bias, const float* scale_dev,
                           int accumulate, void* dx, float* partial, size_t partial_bytes, int* nslices, void* stream,
                           const mi355_bn_bwd_src* bn, const Fp8Extra* f8, const void* acc_mask, int relu) {
  if (nslices) *nslices = 0;
  if (relu && (bn || f8 || partial || accumulate)) MI_FAIL(MI355_EINVAL, "conv_dgrad: the fused ReLU is an inference epilogue of the transposed conv");
  if (acc_mask && (!accumulate || bn || f8)) MI_FAIL(MI355_EINVAL, "conv_dgrad: acc_mask goes with accumulate = 1 on the plain bf16 / fp32 path only");
  if (int e = check_desc(d)) return e;
  if ((d->dtype == MI355_FP8) != (f8 != nullptr)) MI_FAIL(MI355_EINVAL, "fp8 descriptors go through the *_fp8 entry points (and only they)");
  if (f8 && bn) MI_FAIL(MI355_EINVAL, "fp8 dgrad: no BatchNorm-backward epilogue");
  hipStream_t st = as_stream(stream);
  const int s = d->stride;
  const size_t esz = d->dtype == MI355_F32 ? 4 : 2;        // dx element size (bf16 for fp8 operands)
  bool need_zero = false;
  for (int py = 0; py < s && !need_zero; ++py) {
    int cnt = 0; for (int kh = 0; kh < d->kh; ++kh) if ((py + d->pad - kh) % s == 0) ++cnt;
    if (!cnt) need_zero = true;
  }
  for (int px = 0; px < s && !need_zero; ++px) {
    int cnt = 0; for (int kw = 0; kw < d->kw; ++kw) if ((px + d->pad - kw) % s == 0) ++cnt;
    if (!cnt) need_zero = true;
  }
  if (need_zero && !accumulate) {
    const size_t n16 = (size_t)d->N * d->Hi * d->Wi * d->Ci * esz / 16;   // Ci*esz is a multiple of 16
    int grid = (int)((n16 + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint4*>(dx), n16);
    MI_CHECK_LAUNCH("zero_fill");
  }
  GatherArgs a; memset(&a, 0, sizeof(a));
  if (prof_on()) prof_set_tag("dgrad%s k%ds%d %d>%d @%dx%d n%d%s%s%s", f8 ? "8" : "", d->kh, d->stride, d->Ci, d->Co, d->Hi, d->Wi, d->N, partial ? " +stats" : "",
                              accumulate ? (acc_mask ? " +macc" : " +acc") : "", bn ? " +bnb" : "");
  a.A = dy; a.B = wT; a.D = dx; a.bias = bias; a.residual = nullptr; a.scale = scale_dev;
  a.Hi = d->Ho; a.Wi = d->Wo; a.Ci = d->Co;
  a.in_sy = a.in_sx = 1; a.Ho = d->Hi; a.Wo = d->Wi; a.out_sy = a.out_sx = s;
  a.Nout = d->Ci; a.ldd = d->Ci; a.ldb = d->kh * d->kw * d->Co;
  a.accumulate = accumulate ? 1 : 0;
  a.acc_mask = reinterpret_cast<const unsigned char*>(acc_mask);
  a.relu = relu ? 1 : 0;
  int nt = 0;
  for (int py = 0; py < s; ++py)
    for (int px = 0; px < s; ++px) {
      Phase& P = a.ph[a.nphase];
      P.OHp = (d->Hi - py + s - 1) / s; P.OWp = (d->Wi - px + s - 1) / s;
      if (P.OHp <= 0 || P.OWp <= 0) continue;
      P.out_oy = py; P.out_ox = px; P.M = d->N * P.OHp * P.OWp; P.tap0 = nt;
      for (int kh = 0; kh < d->kh; ++kh) {
        if ((py + d->pad - kh) % s != 0) continue;
        for (int kw = 0; kw < d->kw; ++kw) {
          if ((px + d->pad - kw) % s != 0) continue;
          Tap& t = a.taps[nt++]; t.dy = (int8_t)((py + d->pad - kh) / s); t.dx = (int8_t)((px + d->pad - kw) / s);
          t.widx = (int16_t)(kh * d->kw + kw);
        }
      }
      P.ntaps = nt - P.tap0;
      if (P.ntaps == 0) continue;   // region already zeroed (or left untouched when accumulating)
      ++a.nphase;
    }
  if (a.nphase == 0) return MI355_OK;
  static const int merge = getenv("MI355_PHASES") ? atoi(getenv("MI355_PHASES")) : 1;   // 0: one launch per phase (A/B)
  if (merge || a.nphase == 1) {
    // statistics only when every output pixel is produced by this launch (no zero-filled phase)
    if (bn) set_bnb(a, bn, partial, partial_bytes);   // (zero-filled phases carry dy = 0: they add nothing to the sums)
    else if (partial && !need_zero && a.nphase == s * s) { a.stat_partial = partial; a.stat_bytes = partial_bytes; }
    int e;
    if (f8) { a.scale2 = f8->descale_a; a.scale3 = f8->descale_b; a.a_fmt = f8->a_fmt; e = dispatch_gather_fp8(a, st); }
    else e = d->dtype == MI355_BF16 ? dispatch_gather<bf16_t>(a, st) : dispatch_gather<float>(a, st);
    if (nslices) *nslices = a.stat_slices;
    return e;
  }
  if (f8) MI_FAIL(MI355_EINVAL, "fp8 dgrad: MI355_PHASES=0 is a bf16 / fp32 experiment switch");
  const int np = a.nphase;
  for (int i = 0; i < np; ++i) {
    GatherArgs b = a; b.nphase = 1; b.ph[0] = a.ph[i];
    int e = d->dtype == MI355_BF16 ? dispatch_gather<bf16_t>(b, st) : dispatch_gather<float>(b, st);
    if (e) return e;
  }
  return MI355_OK;
}

struct WgradPlan { int S, rows_per_split, nto, nti, ldw, kw3, mt, kw2; };
static int ilog2_exact(int v);
static const int g_wgrad_blocks = getenv("MI355_WG_BLOCKS") ? atoi(getenv("MI355_WG_BLOCKS")) : 768;
static WgradPlan plan_wgrad(const mi355_conv_desc* d) {
  WgradPlan w; w.ldw = d->kh * d->kw * d->Ci;
  const int bkm = d->dtype == MI355_BF16 ? 64 : 32;
  const long M = (long)d->N * d->Ho * d->Wo;
  // 3x3 / stride 1 / pad 1 in bf16 with a power-of-two width: the kw-shared kernel (see wgrad_kw_kernel)
  static const int kw3_on = getenv("MI355_WGRAD_KW") ? atoi(getenv("MI355_WGRAD_KW")) : 1;
  w.kw3 = kw3_on && d->dtype == MI355_BF16 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 &&
          d->Ho == d->Hi && d->Wo == d->Wi &&              // (not a cropped output: the shifted-row trick needs same-size maps)
          d->Wi >= 8 && ilog2_exact(d->Wi) >= 0;
  w.mt = d->Co <= 64 ? 1 : 2;
  // 3x3 / 4x4, stride 2, pad 1 in bf16, output width a power of two in [8, 64]: the parity-image kernel (wgrad_kw2_kernel)
  static const int kw2_on = getenv("MI355_WGRAD_KW2") ? atoi(getenv("MI355_WGRAD_KW2")) : 1;
  w.kw2 = kw2_on && d->dtype == MI355_BF16 && d->kh == d->kw && (d->kh == 3 || d->kh == 4) && d->stride == 2 && d->pad == 1 &&
          d->Hi % 2 == 0 && d->Wi % 2 == 0 && d->Ho == d->Hi / 2 && d->Wo == d->Wi / 2 && d->Wo >= 8 && d->Wo <= 64 &&
          ilog2_exact(d->Wo) >= 0;
  long tiles;
  if (w.kw2) { w.nto = cdiv(d->Co, 64 * w.mt); w.nti = cdiv(d->Ci, 64); tiles = (long)w.nto * d->kh * w.nti; }
  else if (w.kw3) { w.nto = cdiv(d->Co, 64 * w.mt); w.nti = cdiv(d->Ci, 64); tiles = (long)w.nto * 3 * w.nti; }
  else { w.nto = cdiv(d->Co, 128); w.nti = cdiv(w.ldw, 128); tiles = (long)w.nto * w.nti; }
  // split count: fill the chip (3 blocks per CU), but keep >= 16 reduction steps per block while at least one block
  // per CU remains -- short blocks are all prologue / epilogue and every split costs a full fp32 slab write + read.
  // (measured per layer, B=64 @256x256: see DESIGN.md)
  const long ksteps = (M + bkm - 1) / bkm;
  long S = (g_wgrad_blocks + tiles - 1) / tiles;
  long S16 = ksteps / 16, S256 = (256 + tiles - 1) / tiles;
  long lo = S16 > S256 ? S16 : S256;
  if (S > lo) S = lo;
  if (tiles >= 384) S = 1;                       // enough tiles on their own: direct write, no slab pass
  if (S > ksteps) S = ksteps;
  if (S < 1) S = 1;
  long rps = (M + S - 1) / S; rps = ((rps + bkm - 1) / bkm) * bkm;
  S = (M + rps - 1) / rps;
  w.S = (int)S; w.rows_per_split = (int)rps;
  return w;
}

extern "C" size_t mi355_conv_wgrad_workspace(const mi355_conv_desc* d) {
  WgradPlan w = plan_wgrad(d);
  return (size_t)w.S * d->Co * w.ldw * sizeof(float);
}

extern "C" int mi355_conv_wgrad(const mi355_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                                void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_desc(d, true)) return e;
  hipStream_t st = as_stream(stream);
  const int CH = d->dtype == MI355_BF16 ? 8 : 4;
  if (d->Ci % CH || d->Co % CH) MI_FAIL(MI355_EINVAL, "wgrad: channels must be multiples of %d", CH);
  int cshift = ilog2_exact(d->Ci / CH);
  if (cshift < 0) MI_FAIL(MI355_EINVAL, "wgrad: Ci/%d must be a power of two", CH);
  WgradPlan w = plan_wgrad(d);
  const size_t need = (size_t)w.S * d->Co * w.ldw * sizeof(float);
  const bool direct = (w.S == 1 && !accumulate);
  if (!direct && (ws == nullptr || ws_bytes < need)) MI_FAIL(MI355_EWORKSPACE, "wgrad workspace %zu < %zu", ws_bytes, need);
  if (w.kw2) {
    WgradKw2Args k; memset(&k, 0, sizeof(k));
    k.X = x; k.DY = dy; k.out = direct ? dw : reinterpret_cast<float*>(ws);
    k.H = d->Hi; k.W = d->Wi; k.Ho = d->Ho; k.Ci = d->Ci; k.Co = d->Co;
    k.lwo = ilog2_exact(d->Wo); k.lw = k.lwo > 6 ? 6 : k.lwo;
    k.M = d->N * d->Ho * d->Wo; k.rows_per_split = w.rows_per_split; k.ldw = w.ldw;
    k.slab_stride = (long)d->Co * w.ldw; k.nto = w.nto; k.nci = w.nti;
    k.x_bytes = (unsigned)((long)d->N * d->Hi * d->Wi * d->Ci * 2); k.dy_bytes = (unsigned)((long)k.M * d->Co * 2);
    k.dHo = make_fastdiv(d->Ho);
    {
      if (prof_on()) prof_set_tag("wgrad_kw2 k%ds%d %d>%d @%dx%d n%d S%d", d->kh, d->stride, d->Ci, d->Co, d->Hi, d->Wi, d->N, w.S);
      ProfScope ps(st, 2.0 * k.M * (double)d->Co * w.ldw, (double)k.x_bytes + (double)k.dy_bytes + 4.0 * d->Co * w.ldw);
      dim3 grid(w.nto * d->kh * w.nti * w.S);
      if (d->kh == 3) { if (w.mt == 1) hipLaunchKernelGGL((wgrad_kw2_kernel<1, 3>), grid, dim3(256), 0, st, k); else hipLaunchKernelGGL((wgrad_kw2_kernel<2, 3>), grid, dim3(256), 0, st, k); }
      else { if (w.mt == 1) hipLaunchKernelGGL((wgrad_kw2_kernel<1, 4>), grid, dim3(256), 0, st, k); else hipLaunchKernelGGL((wgrad_kw2_kernel<2, 4>), grid, dim3(256), 0, st, k); }
      MI_CHECK_LAUNCH("wgrad_kw2");
    }
    if (!direct) launch_slab_reduce(reinterpret_cast<const float*>(ws), dw, k.slab_stride, w.S, k.slab_stride, accumulate, st);
    return MI355_OK;
  }
  if (w.kw3) {
    WgradKwArgs k; memset(&k, 0, sizeof(k));
    k.X = x; k.DY = dy; k.out = direct ? dw : reinterpret_cast<float*>(ws);
    k.H = d->Hi; k.W = d->Wi; k.Ci = d->Ci; k.Co = d->Co;
    k.lwf = ilog2_exact(d->Wi); k.lw = k.lwf > 6 ? 6 : k.lwf; k.halo = d->Wi > 64;
    k.M = d->N * d->Ho * d->Wo; k.rows_per_split = w.rows_per_split; k.ldw = w.ldw;
    k.slab_stride = (long)d->Co * w.ldw; k.nto = w.nto; k.nci = w.nti;
    k.x_bytes = (unsigned)((long)d->N * d->Hi * d->Wi * d->Ci * 2); k.dy_bytes = (unsigned)((long)k.M * d->Co * 2);
    k.dH = make_fastdiv(d->Hi);
    {
      if (prof_on()) prof_set_tag("wgrad_kw k%ds%d %d>%d @%dx%d n%d S%d", d->kh, d->stride, d->Ci, d->Co, d->Hi, d->Wi, d->N, w.S);
      ProfScope ps(st, 2.0 * k.M * (double)d->Co * w.ldw, (double)k.x_bytes + (double)k.dy_bytes + 4.0 * d->Co * w.ldw);
      dim3 grid(w.nto * 3 * w.nti * w.S);
      if (w.mt == 1) hipLaunchKernelGGL(wgrad_kw_kernel<1>, grid, dim3(256), 0, st, k);
      else hipLaunchKernelGGL(wgrad_kw_kernel<2>, grid, dim3(256), 0, st, k);
      MI_CHECK_LAUNCH("wgrad_kw");
    }
    if (!direct) {
      long n = k.slab_stride;
      launch_slab_reduce(reinterpret_cast<const float*>(ws), dw, n, w.S, k.slab_stride, accumulate, st);
      MI_CHECK_LAUNCH("slab_reduce");
    }
    return MI355_OK;
  }
  WgradArgs a; memset(&a, 0, sizeof(a));
  a.X = x; a.DY = dy; a.out = direct ? dw : reinterpret_cast<float*>(ws);
  a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
  a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.cshift = cshift;
  a.M = d->N * d->Ho * d->Wo; a.rows_per_split = w.rows_per_split; a.ldw = w.ldw;
  a.slab_stride = (long)d->Co * w.ldw; a.nto = w.nto; a.nti = w.nti;
  a.dWo = make_fastdiv(d->Wo); a.dHo = make_fastdiv(d->Ho);
  { const long esz = d->dtype == MI355_BF16 ? 2 : 4; a.x_bytes = (unsigned)((long)d->N * d->Hi * d->Wi * d->Ci * esz); a.dy_bytes = (unsigned)((long)a.M * d->Co * esz); }
  {
    if (prof_on()) prof_set_tag("wgrad k%ds%d %d>%d @%dx%d n%d S%d", d->kh, d->stride, d->Ci, d->Co, d->Hi, d->Wi, d->N, w.S);
    ProfScope ps(st, 2.0 * a.M * (double)d->Co * w.ldw, (double)a.x_bytes + (double)a.dy_bytes + 4.0 * d->Co * w.ldw);
    dim3 grid(w.nto * w.nti * w.S);
    if (d->dtype == MI355_BF16) hipLaunchKernelGGL(wgrad_gemm_kernel<bf16_t>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wgrad_gemm_kernel<float>, grid, dim3(256), 0, st, a);
    MI_CHECK_LAUNCH("wgrad_gemm");
  }
  if (!direct) {
    long n = a.slab_stride;
    launch_slab_reduce(reinterpret_cast<const float*>(ws), dw, n, w.S, a.slab_stride, accumulate, st);
    MI_CHECK_LAUNCH("slab_reduce");
  }
  return MI355_OK;
}


// ------------------------------------------------------------------------------------ grouped weight gradients
// items: HOST array.  Problems the specialised kernels take (3x3 stride 1 -> wgrad_kw, 3x3 / 4x4 stride 2 -> wgrad_kw2), and
// any problem that fills the chip on its own, run through mi355_conv_wgrad one by one; the rest -- the generic kernel's
// small problems -- are launched in groups of up to WG_MAX with a split count chosen for the GROUP.
static bool group_eligible(const mi355_conv_desc* d, const WgradPlan& w) {
  return !w.kw2 && !w.kw3 && (long)w.nto * w.nti < 384;
}
// ... and of those, the ones the 256 x 256-tile kernel takes (wgrad_group256_kernel): bf16, whole 256-wide tiles both ways
static bool group256_eligible(const mi355_conv_desc* d, const WgradPlan& w) {
  static const bool on = !(getenv("MI355_WGRAD_GROUP256") && atoi(getenv("MI355_WGRAD_GROUP256")) == 0);      // A/B switch
  return on && d->dtype == MI355_BF16 && d->kh == 1 && d->kw == 1 && d->pad == 0 && d->Co % 256 == 0 && d->Ci % 256 == 0;
}
struct GroupPlan { int S, rps; size_t ws_off; };
// tile: 128 (wgrad_group_kernel) or 256 (wgrad_group256_kernel)
static long group_plan(const mi355_wgrad_item* items, const int* idx, int n, GroupPlan* gp, size_t* ws_bytes, int tile = 128) {
  // equal work per block: block-steps W = sum tiles_i * ksteps_i; aim at 3 blocks per CU, never fewer than 16 K-steps per block
  // (256-wide tiles: one 8-wave block per CU)
  double W = 0;
  for (int k = 0; k < n; ++k) {
    const mi355_conv_desc* d = &items[idx[k]].d; WgradPlan w = plan_wgrad(d);
    const int bkm = d->dtype == MI355_BF16 ? 64 : 32;
    const long tiles = tile == 256 ? (long)(d->Co / 256) * (w.ldw / 256) : (long)w.nto * w.nti;
    W += (double)tiles * (((long)d->N * d->Ho * d->Wo + bkm - 1) / bkm);
  }
  static const int group_blocks = getenv("MI355_WG_GROUP_BLOCKS") ? atoi(getenv("MI355_WG_GROUP_BLOCKS")) : 512;     // (768 -> 512: -0.09 ms / iteration, three same-box pairs)
  static const int group256_blocks = getenv("MI355_WG_GROUP256_BLOCKS") ? atoi(getenv("MI355_WG_GROUP256_BLOCKS")) : 256;
  long per = (long)(W / (tile == 256 ? group256_blocks : group_blocks)) + 1; if (per < 16) per = 16;
  long blocks = 0; size_t off = 0;
  // 256-wide tiles run one block per CU: a grid of 294 blocks on 256 CUs would take two rounds, the second nearly empty -- lengthen
  // the blocks until the group fits the target (rounding the split counts up is what overshoots it)
  for (int trial = 0; tile == 256 && trial < 64; ++trial) {
    long b = 0;
    for (int k = 0; k < n; ++k) {
      const mi355_conv_desc* d = &items[idx[k]].d; WgradPlan w = plan_wgrad(d);
      const long M = (long)d->N * d->Ho * d->Wo, ksteps = (M + 63) / 64;
      long S = (ksteps + per - 1) / per; if (S < 1) S = 1;
      long rps = (M + S - 1) / S; rps = ((rps + 63) / 64) * 64;
      S = (M + rps - 1) / rps;
      b += (long)(d->Co / 256) * (w.ldw / 256) * S;
    }
    if (b <= group256_blocks) break;
    per += per / 16 + 1;
  }
  for (int k = 0; k < n; ++k) {
    const mi355_wgrad_item& it = items[idx[k]]; const mi355_conv_desc* d = &it.d; WgradPlan w = plan_wgrad(d);
    const int bkm = d->dtype == MI355_BF16 ? 64 : 32;
    const long M = (long)d->N * d->Ho * d->Wo, ksteps = (M + bkm - 1) / bkm;
    long S = (ksteps + per - 1) / per; if (S < 1) S = 1;
    long rps = (M + S - 1) / S; rps = ((rps + bkm - 1) / bkm) * bkm;
    S = (M + rps - 1) / rps;
    gp[k].S = (int)S; gp[k].rps = (int)rps; gp[k].ws_off = off;
    if (S > 1 || it.accumulate) off += (size_t)S * d->Co * w.ldw * sizeof(float);
    blocks += (tile == 256 ? (long)(d->Co / 256) * (w.ldw / 256) : (long)w.nto * w.nti) * S;
  }
  if (ws_bytes) *ws_bytes = off;
  return blocks;
}
// 3x3 / stride-1 problems (wgrad_kw_kernel) small enough to share a launch: fewer than 24 K steps per block at 768 blocks
static bool kw_group_eligible(const mi355_conv_desc* d, const WgradPlan& w) {
  static const bool on = !(getenv("MI355_WGRAD_KW_GROUP") && atoi(getenv("MI355_WGRAD_KW_GROUP")) == 0);      // A/B switch
  if (!on || !w.kw3) return false;
  const long M = (long)d->N * d->Ho * d->Wo, ksteps = (M + 63) / 64;
  return (long)w.nto * 3 * w.nti * ksteps < 768L * 24;
}
static long kw_group_plan(const mi355_wgrad_item* items, const int* idx, int n, GroupPlan* gp, size_t* ws_bytes) {
  double W = 0;
  for (int k = 0; k < n; ++k) {
    const mi355_conv_desc* d = &items[idx[k]].d; WgradPlan w = plan_wgrad(d);
    W += (double)w.nto * 3 * w.nti * (((long)d->N * d->Ho * d->Wo + 63) / 64);
  }
  static const int group_blocks = getenv("MI355_WG_KW_GROUP_BLOCKS") ? atoi(getenv("MI355_WG_KW_GROUP_BLOCKS")) : 512;      // (512 / 768 / 1024: 32.47 / 32.55 / 32.64 ms per iteration, same box)
  long per = (long)(W / group_blocks) + 1; if (per < 16) per = 16;
  long blocks = 0; size_t off = 0;
  for (int k = 0; k < n; ++k) {
    const mi355_wgrad_item& it = items[idx[k]]; const mi355_conv_desc* d = &it.d; WgradPlan w = plan_wgrad(d);
    const long M = (long)d->N * d->Ho * d->Wo, ksteps = (M + 63) / 64;
    long S = (ksteps + per - 1) / per; if (S < 1) S = 1;
    long rps = (M + S - 1) / S; rps = ((rps + 63) / 64) * 64;
    S = (M + rps - 1) / rps;
    gp[k].S = (int)S; gp[k].rps = (int)rps; gp[k].ws_off = off;
    if (S > 1 || it.accumulate) off += (size_t)S * d->Co * w.ldw * sizeof(float);
    blocks += (long)w.nto * 3 * w.nti * S;
  }
  if (ws_bytes) *ws_bytes = off;
  return blocks;
}
static int launch_wgrad_kw_group(const mi355_wgrad_item* items, const int* idx, int n, void* ws, size_t ws_bytes, hipStream_t st) {
  GroupPlan gp[WGK_MAX]; size_t need = 0;
  kw_group_plan(items, idx, n, gp, &need);
  if (need && (!ws || ws_bytes < need)) MI_FAIL(MI355_EWORKSPACE, "wgrad kw group workspace %zu < %zu", ws_bytes, need);
  WgradKwGroupArgs g; memset(&g, 0, sizeof(g));
  SlabGroupArgs sg; memset(&sg, 0, sizeof(sg));
  int nb = 0, nsb = 0;
  double flops = 0, bytes = 0;
  const int mt = plan_wgrad(&items[idx[0]].d).mt;
  for (int k = 0; k < n; ++k) {
    const mi355_wgrad_item& it = items[idx[k]]; const mi355_conv_desc* d = &it.d;
    WgradPlan w = plan_wgrad(d);
    const bool direct = gp[k].S == 1 && !it.accumulate;
    WgradKwArgs& a = g.p[k];
    a.X = it.x; a.DY = it.dy; a.out = direct ? it.dw : reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + gp[k].ws_off);
    a.H = d->Hi; a.W = d->Wi; a.Ci = d->Ci; a.Co = d->Co;
    a.lwf = ilog2_exact(d->Wi); a.lw = a.lwf > 6 ? 6 : a.lwf; a.halo = d->Wi > 64;
    a.M = d->N * d->Ho * d->Wo; a.rows_per_split = gp[k].rps; a.ldw = w.ldw;
    a.slab_stride = (long)d->Co * w.ldw; a.nto = w.nto; a.nci = w.nti;
    a.x_bytes = (unsigned)((long)d->N * d->Hi * d->Wi * d->Ci * 2); a.dy_bytes = (unsigned)((long)a.M * d->Co * 2);
    a.dH = make_fastdiv(d->Hi);
    g.bstart[k] = nb; nb += w.nto * 3 * w.nti * gp[k].S;
    flops += 2.0 * a.M * (double)d->Co * w.ldw; bytes += (double)a.x_bytes + a.dy_bytes + 4.0 * d->Co * w.ldw;
    if (!direct) {
      SlabItem& q = sg.it[sg.n];
      q.slabs = a.out; q.out = it.dw; q.n4 = a.slab_stride / 4; q.stride = a.slab_stride; q.S = gp[k].S; q.accumulate = it.accumulate;
      sg.bstart[sg.n] = nsb; nsb += cdiv(q.n4, 256); ++sg.n;
    }
  }
  g.bstart[n] = nb; g.n = n; sg.bstart[sg.n] = nsb;
  {
    if (prof_on()) {
      const mi355_conv_desc* d0 = &items[idx[0]].d;
      prof_set_tag("wgrad_kw_group x%d blocks%d first k%ds%d %d>%d @%dx%d", n, nb, d0->kh, d0->stride, d0->Ci, d0->Co, d0->Hi, d0->Wi);
    }
    ProfScope ps(st, flops, bytes);
    if (mt == 1) hipLaunchKernelGGL(wgrad_kw_group_kernel<1>, dim3(nb), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(wgrad_kw_group_kernel<2>, dim3(nb), dim3(256), 0, st, g);
    MI_CHECK_LAUNCH("wgrad_kw_group");
  }
  if (sg.n) {
    double sb = 0; for (int k = 0; k < sg.n; ++k) sb += 16.0 * sg.it[k].n4 * (sg.it[k].S + 1);
    char lab[64];
    if (prof_on()) snprintf(lab, sizeof(lab), "slab_reduce_group x%d", sg.n);
    ProfScope ps(st, 0.0, sb, 2, prof_on() ? lab : nullptr);
    hipLaunchKernelGGL(slab_reduce_group_kernel, dim3(nsb), dim3(256), 0, st, sg); MI_CHECK_LAUNCH("slab_reduce_group");
  }
  return MI355_OK;
}
// The launches mi355_conv_wgrad_grouped makes of `items`, in order, handed to `fn(kind, idx, m)`: kind 0 = one item through
// mi355_conv_wgrad, 1 = a group of the generic kernel, 2 = a group of the 3x3 / stride-1 kernel, 3 = a group of the 256 x 256-tile kernel.  One walk for the launcher and
// for the workspace size, so the two cannot disagree.  Two items that write the same dw (one conv used twice in a backward:
// overwrite, then accumulate) must neither share a launch -- the overwrite and the read-modify-write would race -- nor change
// their order: whatever is pending goes first.
template <typename F>
static int walk_wgrad_groups(const mi355_wgrad_item* items, int n, F&& fn) {
  int gi[WG_MAX], ki[WGK_MAX], bi[WG_MAX]; int gm = 0, km = 0, bm = 0;
  auto flush_g = [&]() -> int { if (!gm) return 0; int e = fn(1, gi, gm); gm = 0; return e; };
  auto flush_k = [&]() -> int { if (!km) return 0; int e = fn(2, ki, km); km = 0; return e; };
  auto flush_b = [&]() -> int { if (!bm) return 0; int e = fn(3, bi, bm); bm = 0; return e; };
  for (int i = 0; i < n; ++i) {
    const mi355_wgrad_item& it = items[i];
    WgradPlan w = plan_wgrad(&it.d);
    bool shares = false;
    for (int k = 0; k < gm; ++k) shares = shares || items[gi[k]].dw == it.dw;
    for (int k = 0; k < km; ++k) shares = shares || items[ki[k]].dw == it.dw;
    for (int k = 0; k < bm; ++k) shares = shares || items[bi[k]].dw == it.dw;
    if (shares) { if (int e = flush_g()) return e; if (int e = flush_k()) return e; if (int e = flush_b()) return e; }
    if (kw_group_eligible(&it.d, w)) {
      if (km && (km == WGK_MAX || plan_wgrad(&items[ki[0]].d).mt != w.mt)) { if (int e = flush_k()) return e; }
      ki[km++] = i;
    } else if (group_eligible(&it.d, w) && group256_eligible(&it.d, w)) {
      if (bm == WG_MAX) { if (int e = flush_b()) return e; }
      bi[bm++] = i;
    } else if (group_eligible(&it.d, w)) {
      if (gm && (gm == WG_MAX || items[gi[0]].d.dtype != it.d.dtype)) { if (int e = flush_g()) return e; }
      gi[gm++] = i;
    } else {
      int one = i;
      if (int e = fn(0, &one, 1)) return e;
    }
  }
  if (int e = flush_g()) return e;
  if (int e = flush_b()) return e;
  return flush_k();
}
extern "C" size_t mi355_conv_wgrad_grouped_workspace(const mi355_wgrad_item* items, int n) {
  if (!items || n < 1) return 0;
  size_t need = 0;
  (void)walk_wgrad_groups(items, n, [&](int kind, const int* idx, int m) -> int {
    size_t b = 0;
    if (kind == 0) b = mi355_conv_wgrad_workspace(&items[idx[0]].d);
    else if (kind == 1) { GroupPlan gp[WG_MAX]; group_plan(items, idx, m, gp, &b); }
    else if (kind == 3) { GroupPlan gp[WG_MAX]; group_plan(items, idx, m, gp, &b, 256); }
    else { GroupPlan gp[WGK_MAX]; kw_group_plan(items, idx, m, gp, &b); }
    if (b > need) need = b;
    return 0;
  });
  return need;
}
static int launch_wgrad_group(const mi355_wgrad_item* items, const int* idx, int n, void* ws, size_t ws_bytes, hipStream_t st, int tile = 128) {
  GroupPlan gp[WG_MAX]; size_t need = 0;
  group_plan(items, idx, n, gp, &need, tile);
  if (need && (!ws || ws_bytes < need)) MI_FAIL(MI355_EWORKSPACE, "wgrad group workspace %zu < %zu", ws_bytes, need);
  WgradGroupArgs g; memset(&g, 0, sizeof(g));
  SlabGroupArgs sg; memset(&sg, 0, sizeof(sg));
  int nb = 0, nsb = 0;
  for (int k = 0; k < n; ++k) {
    const mi355_wgrad_item& it = items[idx[k]]; const mi355_conv_desc* d = &it.d;
    if (int e = check_desc(d)) return e;
    const int CH = d->dtype == MI355_BF16 ? 8 : 4;
    if (d->Ci % CH || d->Co % CH) MI_FAIL(MI355_EINVAL, "wgrad: channels must be multiples of %d", CH);
    const int cshift = ilog2_exact(d->Ci / CH);
    if (cshift < 0) MI_FAIL(MI355_EINVAL, "wgrad: Ci/%d must be a power of two", CH);
    WgradPlan w = plan_wgrad(d);
    const bool direct = gp[k].S == 1 && !it.accumulate;
    WgradArgs& a = g.p[k];
    a.X = it.x; a.DY = it.dy; a.out = direct ? it.dw : reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + gp[k].ws_off);
    a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.cshift = cshift;
    a.M = d->N * d->Ho * d->Wo; a.rows_per_split = gp[k].rps; a.ldw = w.ldw;
    a.slab_stride = (long)d->Co * w.ldw; a.nto = w.nto; a.nti = w.nti;
    if (tile == 256) { a.nto = d->Co / 256; a.nti = w.ldw / 256; }
    a.dWo = make_fastdiv(d->Wo); a.dHo = make_fastdiv(d->Ho);
    const long esz = d->dtype == MI355_BF16 ? 2 : 4;
    a.x_bytes = (unsigned)((long)d->N * d->Hi * d->Wi * d->Ci * esz); a.dy_bytes = (unsigned)((long)a.M * d->Co * esz);
    g.bstart[k] = nb; nb += a.nto * a.nti * gp[k].S;
    if (!direct) {
      SlabItem& q = sg.it[sg.n];
      q.slabs = a.out; q.out = it.dw; q.n4 = a.slab_stride / 4; q.stride = a.slab_stride; q.S = gp[k].S; q.accumulate = it.accumulate;
      sg.bstart[sg.n] = nsb; nsb += cdiv(q.n4, 256); ++sg.n;
    }
  }
  g.bstart[n] = nb; g.n = n; sg.bstart[sg.n] = nsb;
  {
    double flops = 0, bytes = 0;
    for (int k = 0; k < n; ++k) { const WgradArgs& a = g.p[k]; flops += 2.0 * a.M * (double)a.Co * a.ldw; bytes += (double)a.x_bytes + a.dy_bytes + 4.0 * a.Co * a.ldw; }
    if (prof_on()) {
      const mi355_conv_desc* d0 = &items[idx[0]].d;
      prof_set_tag("wgrad_group%s x%d blocks%d first k%ds%d %d>%d @%dx%d", tile == 256 ? "256" : "", n, nb, d0->kh, d0->stride, d0->Ci, d0->Co, d0->Hi, d0->Wi);
    }
    ProfScope ps(st, flops, bytes);
    if (tile == 256) {
      static bool attr_set = false;
      if (!attr_set) { (void)hipFuncSetAttribute((const void*)wgrad_group256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG256_SMEM); attr_set = true; }
      hipLaunchKernelGGL(wgrad_group256_kernel, dim3(nb), dim3(512), WG256_SMEM, st, g);
    }
    else if (items[idx[0]].d.dtype == MI355_BF16) hipLaunchKernelGGL(wgrad_group_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(wgrad_group_kernel<float>, dim3(nb), dim3(256), 0, st, g);
    MI_CHECK_LAUNCH("wgrad_group");
  }
  if (sg.n) {
    double sb = 0; for (int k = 0; k < sg.n; ++k) sb += 16.0 * sg.it[k].n4 * (sg.it[k].S + 1);
    char lab[64];
    if (prof_on()) snprintf(lab, sizeof(lab), "slab_reduce_group x%d", sg.n);
    ProfScope ps(st, 0.0, sb, 2, prof_on() ? lab : nullptr);
    hipLaunchKernelGGL(slab_reduce_group_kernel, dim3(nsb), dim3(256), 0, st, sg); MI_CHECK_LAUNCH("slab_reduce_group");
  }
  return MI355_OK;
}
extern "C" int mi355_conv_wgrad_grouped(const mi355_wgrad_item* items, int n, void* ws, size_t ws_bytes, void* stream) {
  if (!items || n < 1) MI_FAIL(MI355_EINVAL, "wgrad_grouped: no items");
  hipStream_t st = as_stream(stream);
  for (int i = 0; i < n; ++i) {
    if (int e = check_desc(&items[i].d)) return e;
    if (!items[i].x || !items[i].dy || !items[i].dw) MI_FAIL(MI355_EINVAL, "wgrad_grouped: item %d has a null operand", i);
  }
  return walk_wgrad_groups(items, n, [&](int kind, const int* idx, int m) -> int {
    if (kind == 0) { const mi355_wgrad_item& it = items[idx[0]]; return mi355_conv_wgrad(&it.d, it.x, it.dy, it.dw, it.accumulate, ws, ws_bytes, stream); }
    if (kind == 1) return launch_wgrad_group(items, idx, m, ws, ws_bytes, st);
    if (kind == 3) return launch_wgrad_group(items, idx, m, ws, ws_bytes, st, 256);
    return launch_wgrad_kw_group(items, idx, m, ws, ws_bytes, st);
  });
}

// ------------------------------------------------------------------------------------ weight packing
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wt, int O, int Tt, int I, int Ipad) {
  // tile 32(o) x 32(i) per tap through LDS so both the read ([O][T][I]) and the transposed write ([Ipad][T][O]) coalesce;
  // channels I..Ipad-1 of the packed copies are zero (stem: 3 -> 8)
  __shared__ float tile[32][33];
  const int tap = blockIdx.z, o0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int o = o0 + r, i = i0 + tx;
    float v = (o < O && i < I) ? w[((size_t)o * Tt + tap) * I + i] : 0.f;
    tile[r][tx] = v;
    if (wf && o < O && i < Ipad) Elem<T>::st(wf + ((size_t)o * Tt + tap) * Ipad + i, v);
  }
  __syncthreads();
  if (wt)
    for (int r = ty; r < 32; r += 8) {
      int i = i0 + r, o = o0 + tx;
      if (o < O && i < Ipad) Elem<T>::st(wt + ((size_t)i * Tt + tap) * O + o, tile[tx][r]);
    }
}

// All conv weights of one optimizer group in ONE launch (after the optimizer step): items live in device memory,
// block b finds its item by binary search over the items' first-block prefix.
template <typename T>
__global__ void pack_weights_batched_kernel(const mi355_pack_item* __restrict__ items, int nitems) {
  __shared__ float tile[32][33];
  int lo = 0, hi = nitems - 1;
  const int b = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].blk0 <= b) lo = mid; else hi = mid - 1; }
  const mi355_pack_item it = items[lo];
  const int O = it.O, Tt = it.T, I = it.I, Ipad = it.Ipad;
  const int nbi = (Ipad + 31) / 32, nbo = (O + 31) / 32;
  int q = b - it.blk0;
  const int bi = q % nbi; q /= nbi;
  const int bo = q % nbo; const int tap = q / nbo;
  const float* __restrict__ w = it.w;
  T* __restrict__ wf = reinterpret_cast<T*>(it.wf);
  T* __restrict__ wt = reinterpret_cast<T*>(it.wt);
  const int o0 = bo * 32, i0 = bi * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    int o = o0 + r, i = i0 + tx;
    float v = (o < O && i < I) ? w[((size_t)o * Tt + tap) * I + i] : 0.f;
    tile[r][tx] = v;
    if (wf && o < O && i < Ipad) Elem<T>::st(wf + ((size_t)o * Tt + tap) * Ipad + i, v);
  }
  __syncthreads();
  if (wt)
    for (int r = ty; r < 32; r += 8) {
      int i = i0 + r, o = o0 + tx;
      if (o < O && i < Ipad) Elem<T>::st(wt + ((size_t)i * Tt + tap) * O + o, tile[tx][r]);
    }
}
extern "C" int mi355_pack_weights_batched(const mi355_pack_item* items_dev, int nitems, int total_blocks, int dtype, void* stream) {
  if (!items_dev || nitems < 1 || total_blocks < 1) MI_FAIL(MI355_EINVAL, "pack_weights_batched: bad args");
  if (dtype == MI355_BF16) hipLaunchKernelGGL(pack_weights_batched_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, as_stream(stream), items_dev, nitems);
  else if (dtype == MI355_F32) hipLaunchKernelGGL(pack_weights_batched_kernel<float>, dim3(total_blocks), dim3(256), 0, as_stream(stream), items_dev, nitems);
  else MI_FAIL(MI355_EINVAL, "bad dtype");
  MI_CHECK_LAUNCH("pack_weights_batched");
  return MI355_OK;
}

extern "C" int mi355_pack_weights(const float* w, void* wf, void* wt, int O, int Tt, int I, int Ipad, int dtype, void* stream) {
  if (!w || O < 1 || Tt < 1 || I < 1 || Ipad < I) MI_FAIL(MI355_EINVAL, "pack_weights: bad args");
  dim3 grid(cdiv(Ipad, 32), cdiv(O, 32), Tt);
  if (dtype == MI355_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), w, (bf16_t*)wf, (bf16_t*)wt, O, Tt, I, Ipad);
  else if (dtype == MI355_F32) hipLaunchKernelGGL(pack_weights_kernel<float>, grid, dim3(256), 0, as_stream(stream), w, (float*)wf, (float*)wt, O, Tt, I, Ipad);
  else MI_FAIL(MI355_EINVAL, "bad dtype");
  MI_CHECK_LAUNCH("pack_weights");
  return MI355_OK;
}
