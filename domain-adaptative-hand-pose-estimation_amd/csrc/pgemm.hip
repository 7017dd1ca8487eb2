// Persistent, wave-specialised pipelined GEMM for the 1x1 / unit-stride convolutions (forward and input gradient), bf16, gfx950.
//
//   D[m][n] = epilogue( sum_k A[m][k] * B[n][k] )      A: [M][K] activations (NHWC rows ARE the GEMM rows: no gather),
//                                                      B: [N][K] packed weights, K = channels (multiple of 64, >= 128)
//
// Why a second kernel beside gather_gemm_kernel (igemm.hip).  A ResNet-50 iteration launches ~190 of these layers with
// M = 4 K .. 16 K rows and K = 128 .. 2048: two to thirty-two 64-deep K steps per tile.  Measured on the general kernel and on a
// first persistent build of this one (profiles/r03_pgemm_phases.txt: the same launch with its output stores, its MFMA step or its
// LDS-DMA switched off): the phases of a tile -- operand fetch, MFMA, convert + stage + store (+ BatchNorm statistics) -- do not
// overlap; each adds its full cost, and the instruction overhead of a tile alone (no loads, no MFMA, no stores) is 40 % of the
// launch.  256 -> 1024 @16x16 (B=64): 21 us = skeleton 8.7 + LDS-DMA 3.9 + MFMA 5.3 + stores 3.9.  So the phases are made to
// run CONCURRENTLY, inside one 512-thread block per CU:
//   * waves 0-3 (one per SIMD) are the MFMA waves: they own the LDS ring (NS stages of [BM + BN] x 128-byte rows, filled by
//     LDS-DMA with NS-1 K steps in flight, counted vmcnt, ONE barrier per K step, operand addresses = a per-lane offset fixed
//     for the tile + k * 128 in the instruction's scalar offset) and the accumulators; when a tile's last K step is done they
//     convert it ((acc + bias) * scale -> bf16) into one of two LDS staging tiles and go straight on to the next tile -- the ring
//     never drains, the first K steps of the next tile are already in flight;
//   * waves 4-7 (their SIMD partners) are the EPILOGUE waves: while the MFMA waves multiply tile i + 1 they stream tile i out of
//     its staging tile -- residual / accumulate (+ bit mask) / ReLU, BatchNorm statistics of the output, coalesced 16-byte
//     stores -- a share per K step, on the VALU, LDS-read and store paths the MFMA waves leave idle.  Their loads and stores
//     have their own vmcnt, so the MFMA waves' counted waits stay exact.
//   All eight waves meet at the one barrier per K step; the schedule is static (no flags, no polling):
//       tile i staged before barrier (i+1)*nk | drained in steps (i+1)*nk .. (i+2)*nk-2 | statistics scratch in step (i+2)*nk-1
//       | folded in step (i+2)*nk | its staging tile rewritten (tile i+2) at the end of step (i+3)*nk-1      [nk = K / 64 >= 2]
// Fragment layout, LDS row swizzle and MFMA operand order are those of gather_gemm_kernel (128-byte rows, chunk ^ (row >> 1) & 7,
// D^T accumulators); both kernels produce the same bits for the same element.
#include "common.h"
#include <stdlib.h>

#include "igemm_common.h"

template <int BM, int BN, int NS>
struct PgSmem {
  static constexpr int kStage = (BM + BN) * 128;
  static constexpr int kOut = BM * BN * 2;                   // one staging tile (bf16, swizzled, no padding)
  static constexpr int kRing = NS * kStage;
  static constexpr int kBytes = kRing + 2 * kOut;
  static_assert(kBytes <= 160 * 1024, "ring + two staging tiles must fit the CU's LDS");
};

// swizzled byte offset of 16-byte chunk c of output-tile row r in a staging tile (rows of BN * 2 bytes, no padding):
// 256-byte rows cover all 64 banks -> 16 rows need 16 chunk positions; 128-byte rows cover half -> row pairs alternate halves
template <int BN> __device__ __forceinline__ int out_swz(int r, int c) {
  if constexpr (BN == 128) return r * 256 + ((c ^ (r & 15)) << 4);
  else return r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
}

// LDS accesses of this kernel are written as inline asm on purpose.  hipcc tracks LDS-DMA (buffer_load ... lds) as a pending
// write to "some LDS address" and, lacking alias information, puts `s_waitcnt vmcnt(0)` in front of every ds_read / ds_write it
// can see -- which would drain the ring (all K steps in flight) before each fragment read and each epilogue store.  The asm
// forms are invisible to that pass; the waits they need are placed by hand (counted vmcnt before the barrier that publishes a
// stage; lgkmcnt through wait statements that name the registers they guard, so no consumer is scheduled above them).
typedef __attribute__((address_space(3))) char* lds_cptr;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(lds_cptr)(const_cast<void*>(p)); }
__device__ __forceinline__ bf16x8_t lds_read16(unsigned a) { bf16x8_t v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); return v; }
__device__ __forceinline__ u32x4_t lds_read16u(unsigned a) { u32x4_t v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); return v; }
__device__ __forceinline__ void lds_write8(unsigned a, uint2 v) { asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write4(unsigned a, float v) { asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ float lds_read4(unsigned a) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); return v; }
#define LDS_WAIT_ALL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Workgroup barrier WITHOUT the fences of __syncthreads(): its workgroup-scope release makes hipcc wait for vmcnt(0), i.e. for
// every LDS-DMA stage in flight.  All cross-wave traffic of this kernel is LDS traffic issued by the asm forms above and drained
// by hand (lgkmcnt / counted vmcnt) before the barrier that publishes it; nothing is handed over through global memory.
#define RAW_BARRIER() asm volatile("s_barrier" ::: "memory")
// diagnostic builds of a launch (MI355_PG_DEBUG & 8): cycle stamps of block 0's wave 0 (MFMA role) and wave 4 (epilogue role)
// go to a buffer of their own ([role][step][8] x 64-bit); nothing else reads them (guide section 7, in-kernel stamps)
#define PG_STAMP(role, step, slot) do { if (stamp && (step) < 256) stamp[((role) * 256 + (step)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)

template <int BM, int BN, int NS, int EPI>
__global__ __launch_bounds__(512) void pgemm_kernel(const GatherArgs p) {
  constexpr int CH = 8;
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int PA = BM / 32, PB = BN / 32;                  // 1-KiB LDS-DMA pieces per MFMA wave and stage
  constexpr int DIST = NS - 1;                               // K steps in flight
  constexpr int CPR = BN / CH;                               // 16-byte chunks per output row
  constexpr int IT = BM * CPR / 256;                         // chunks per epilogue thread and tile
  static_assert(BM * CPR % 256 == 0, "whole passes only");
  using SM = PgSmem<BM, BN, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned smem_base = lds_addr(smem);
  const int M = p.ph[0].M, K = p.Ci, nk = K >> 6, ntn = p.ntn;
  const int dbg = p.hw;        // diagnostic launches only (MI355_PG_DEBUG): 1 no output stores, 2 no MFMA step, 4 no LDS-DMA
  const int G = (int)gridDim.x;
  const int pos = xcd_remap((int)blockIdx.x, G);             // blocks of one XCD take neighbouring tiles (same A rows)
  const int my_tiles = pos < p.ntiles ? (p.ntiles - pos + G - 1) / G : 0;
  const int total = my_tiles * nk;
  if (total == 0) return;
  unsigned long long* stamp = ((dbg & 8) && blockIdx.x == 0 && (t == 0 || t == 256)) ? reinterpret_cast<unsigned long long*>(p.bnb_partial) : nullptr;

  if (wave_u < 4) {
    // ============================================================================================== MFMA waves
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int lc = t & 7, lr = t >> 3;
    const int lcs = lc ^ ((lr >> 1) & 7);                    // the swizzle is applied on the SOURCE address
    const int r31 = lane & 31, hi = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);

    // ---- issue side: tile / K step of the next LDS-DMA stage to launch
    int is_tile = 0, is_kt = 0, is_gs = 0;
    int voffA[PA], voffB[PB];
    auto set_issue_tile = [&](int i) {
      const int lin = i * G + pos, tm = lin / ntn, tn = lin - tm * ntn;
#pragma unroll
      for (int j = 0; j < PA; ++j) { const int m = tm * BM + j * 32 + lr; voffA[j] = m < M ? (m * K + lcs * CH) * 2 : OOB_OFF; }
#pragma unroll
      for (int j = 0; j < PB; ++j) { const int n = tn * BN + j * 32 + lr; voffB[j] = n < p.Nout ? (n * p.ldb + lcs * CH) * 2 : OOB_OFF; }
    };
    auto issue = [&]() {                                     // stage is_gs -> ring slot is_gs % NS
      const int slot = is_gs % NS;
      const int soff = is_kt * 128;
      (void)slot; (void)soff;
#if defined(__HIP_DEVICE_COMPILE__)
      if (!(dbg & 4)) {
        typedef __attribute__((address_space(3))) void* ldsp;
        char* sa = smem + slot * SM::kStage + wave_u * 1024;
        char* sb = sa + BM * 128;
#pragma unroll
        for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsp)(sa + j * 4096), 16, voffA[j], soff, 0, 0);
#pragma unroll
        for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsp)(sb + j * 4096), 16, voffB[j], soff, 0, 0);
      }
#endif
      ++is_gs;
      if (++is_kt == nk) { is_kt = 0; ++is_tile; if (is_tile < my_tiles) set_issue_tile(is_tile); }
    };

    f32x16_t acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    // fragment byte offsets inside a stage (fixed per lane): row part + the swizzled chunk of every 16-deep sub-step
    unsigned fa[MT][4], fb[NT][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i][s] = (unsigned)swz128(wm0 + i * 32 + r31, 2 * s + hi);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j][s] = (unsigned)(BM * 128 + swz128(wn0 + j * 32 + r31, 2 * s + hi));
    }
    // One K step = ONE asm statement: 4 x (MT + NT) fragment reads, software-pipelined one 16-deep sub-step ahead of the 4 x MT x NT
    // MFMAs (two fragment buffers), waits counted in lgkmcnt.  hipcc would sink the MFMAs below the later waits and shuttle the
    // accumulators between the two register files around the conditional epilogue; here they stay in the accumulator file.
#define PG_MFMA(ACC, B, A) "v_mfma_f32_32x32x16_bf16 " ACC ", " B ", " A ", " ACC "\n"
#define PG_RD(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n"
    auto compute = [&](unsigned stage_addr) {
      unsigned xa[MT][4], xb[NT][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < MT; ++i) xa[i][s] = stage_addr + fa[i][s];
#pragma unroll
        for (int j = 0; j < NT; ++j) xb[j][s] = stage_addr + fb[j][s];
      }
      if constexpr (MT == 2 && NT == 2) {
        bf16x8_t a0, a1, b0, b1, c0, c1, d0, d1;      // buffer 0: a0 a1 (A rows) b0 b1 (B rows); buffer 1: c0 c1 / d0 d1
        asm volatile(
            PG_RD("%4", "%12") PG_RD("%6", "%20") PG_RD("%5", "%16") PG_RD("%7", "%24")
            PG_RD("%8", "%13") PG_RD("%10", "%21") PG_RD("%9", "%17") PG_RD("%11", "%25")
            "s_waitcnt lgkmcnt(4)\n"
            PG_MFMA("%0", "%6", "%4") PG_MFMA("%1", "%7", "%4") PG_MFMA("%2", "%6", "%5") PG_MFMA("%3", "%7", "%5")
            PG_RD("%4", "%14") PG_RD("%6", "%22") PG_RD("%5", "%18") PG_RD("%7", "%26")
            "s_waitcnt lgkmcnt(4)\n"
            PG_MFMA("%0", "%10", "%8") PG_MFMA("%1", "%11", "%8") PG_MFMA("%2", "%10", "%9") PG_MFMA("%3", "%11", "%9")
            PG_RD("%8", "%15") PG_RD("%10", "%23") PG_RD("%9", "%19") PG_RD("%11", "%27")
            "s_waitcnt lgkmcnt(4)\n"
            PG_MFMA("%0", "%6", "%4") PG_MFMA("%1", "%7", "%4") PG_MFMA("%2", "%6", "%5") PG_MFMA("%3", "%7", "%5")
            "s_waitcnt lgkmcnt(0)\n"
            PG_MFMA("%0", "%10", "%8") PG_MFMA("%1", "%11", "%8") PG_MFMA("%2", "%10", "%9") PG_MFMA("%3", "%11", "%9")
            : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]),
              "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(c0), "=&v"(c1), "=&v"(d0), "=&v"(d1)
            : "v"(xa[0][0]), "v"(xa[0][1]), "v"(xa[0][2]), "v"(xa[0][3]), "v"(xa[1][0]), "v"(xa[1][1]), "v"(xa[1][2]), "v"(xa[1][3]),
              "v"(xb[0][0]), "v"(xb[0][1]), "v"(xb[0][2]), "v"(xb[0][3]), "v"(xb[1][0]), "v"(xb[1][1]), "v"(xb[1][2]), "v"(xb[1][3])
            : "memory");
      } else if constexpr (MT == 1 && NT == 2) {
        bf16x8_t a0, b0, b1, c0, d0, d1;
        asm volatile(
            PG_RD("%2", "%8") PG_RD("%3", "%12") PG_RD("%4", "%16")
            PG_RD("%5", "%9") PG_RD("%6", "%13") PG_RD("%7", "%17")
            "s_waitcnt lgkmcnt(3)\n"
            PG_MFMA("%0", "%3", "%2") PG_MFMA("%1", "%4", "%2")
            PG_RD("%2", "%10") PG_RD("%3", "%14") PG_RD("%4", "%18")
            "s_waitcnt lgkmcnt(3)\n"
            PG_MFMA("%0", "%6", "%5") PG_MFMA("%1", "%7", "%5")
            PG_RD("%5", "%11") PG_RD("%6", "%15") PG_RD("%7", "%19")
            "s_waitcnt lgkmcnt(3)\n"
            PG_MFMA("%0", "%3", "%2") PG_MFMA("%1", "%4", "%2")
            "s_waitcnt lgkmcnt(0)\n"
            PG_MFMA("%0", "%6", "%5") PG_MFMA("%1", "%7", "%5")
            : "+v"(acc[0][0]), "+v"(acc[0][1]), "=&v"(a0), "=&v"(b0), "=&v"(b1), "=&v"(c0), "=&v"(d0), "=&v"(d1)
            : "v"(xa[0][0]), "v"(xa[0][1]), "v"(xa[0][2]), "v"(xa[0][3]),
              "v"(xb[0][0]), "v"(xb[0][1]), "v"(xb[0][2]), "v"(xb[0][3]), "v"(xb[1][0]), "v"(xb[1][1]), "v"(xb[1][2]), "v"(xb[1][3])
            : "memory");
      } else if constexpr (MT == 2 && NT == 1) {
        bf16x8_t a0, a1, b0, c0, c1, d0;
        asm volatile(
            PG_RD("%2", "%8") PG_RD("%3", "%12") PG_RD("%4", "%16")
            PG_RD("%5", "%9") PG_RD("%6", "%13") PG_RD("%7", "%17")
            "s_waitcnt lgkmcnt(3)\n"
            PG_MFMA("%0", "%4", "%2") PG_MFMA("%1", "%4", "%3")
            PG_RD("%2", "%10") PG_RD("%3", "%14") PG_RD("%4", "%18")
            "s_waitcnt lgkmcnt(3)\n"
            PG_MFMA("%0", "%7", "%5") PG_MFMA("%1", "%7", "%6")
            PG_RD("%5", "%11") PG_RD("%6", "%15") PG_RD("%7", "%19")
            "s_waitcnt lgkmcnt(3)\n"
            PG_MFMA("%0", "%4", "%2") PG_MFMA("%1", "%4", "%3")
            "s_waitcnt lgkmcnt(0)\n"
            PG_MFMA("%0", "%7", "%5") PG_MFMA("%1", "%7", "%6")
            : "+v"(acc[0][0]), "+v"(acc[1][0]), "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(c0), "=&v"(c1), "=&v"(d0)
            : "v"(xa[0][0]), "v"(xa[0][1]), "v"(xa[0][2]), "v"(xa[0][3]), "v"(xa[1][0]), "v"(xa[1][1]), "v"(xa[1][2]), "v"(xa[1][3]),
              "v"(xb[0][0]), "v"(xb[0][1]), "v"(xb[0][2]), "v"(xb[0][3])
            : "memory");
      } else {
        bf16x8_t a0, b0, c0, d0;
        asm volatile(
            PG_RD("%1", "%5") PG_RD("%2", "%9") PG_RD("%3", "%6") PG_RD("%4", "%10")
            "s_waitcnt lgkmcnt(2)\n"
            PG_MFMA("%0", "%2", "%1")
            PG_RD("%1", "%7") PG_RD("%2", "%11")
            "s_waitcnt lgkmcnt(2)\n"
            PG_MFMA("%0", "%4", "%3")
            PG_RD("%3", "%8") PG_RD("%4", "%12")
            "s_waitcnt lgkmcnt(2)\n"
            PG_MFMA("%0", "%2", "%1")
            "s_waitcnt lgkmcnt(0)\n"
            PG_MFMA("%0", "%4", "%3")
            : "+v"(acc[0][0]), "=&v"(a0), "=&v"(b0), "=&v"(c0), "=&v"(d0)
            : "v"(xa[0][0]), "v"(xa[0][1]), "v"(xa[0][2]), "v"(xa[0][3]), "v"(xb[0][0]), "v"(xb[0][1]), "v"(xb[0][2]), "v"(xb[0][3])
            : "memory");
      }
    };
#undef PG_MFMA
#undef PG_RD

    // ---- hand-over of a finished tile: (acc + bias) * scale -> bf16 -> staging tile `outs`
    const float scale = p.scale ? *p.scale : 1.0f;
    auto stage_out = [&](unsigned outs, int n0) {
      float4 bq[NT][4];                                      // bias of this lane's 4-wide column runs, fetched up front
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn0 + j * 32 + 8 * g + 4 * hi;  // Nout is a multiple of 8: a run is inside or outside as a whole
          bq[j][g] = (p.bias && n < p.Nout) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // the last MFMAs' results before hipcc's reads of them (asm is opaque to its hazard pass)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ml = wm0 + i * 32 + r31, nl = wn0 + j * 32 + 8 * g + 4 * hi;
            const float bb[4] = {bq[j][g].x, bq[j][g].y, bq[j][g].z, bq[j][g].w};
            union { bf16_t h[4]; uint2 q; } u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u.h[e] = (bf16_t)((acc[i][j][4 * g + e] + bb[e]) * scale);
            lds_write8(outs + out_swz<BN>(ml, nl >> 3) + (nl & 7) * 2, u.q);
          }
      LDS_WAIT_ALL();                                        // ... published by the next barrier
    };

    // ---- the pipeline
    set_issue_tile(0);
#pragma unroll
    for (int d = 0; d < DIST; ++d)
      if (is_gs < total) issue();
    int kt = 0, tile_i = 0;
    zero_acc();
    for (int gs = 0; gs < total; ++gs) {
      PG_STAMP(0, gs, 0);
      // this wave's pieces of stage gs have landed once at most the pieces of the DIST-1 younger stages are outstanding
      // (the MFMA waves issue nothing else that counts in vmcnt except the bias loads of stage_out, which hipcc waits for itself)
#if defined(__HIP_DEVICE_COMPILE__)
      if (gs + DIST - 1 < total && DIST > 1) {
        if constexpr ((DIST - 1) * (PA + PB) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr ((DIST - 1) * (PA + PB) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr ((DIST - 1) * (PA + PB) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr ((DIST - 1) * (PA + PB) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr ((DIST - 1) * (PA + PB) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#endif
      PG_STAMP(0, gs, 1);
      RAW_BARRIER();                                         // barrier gs: everyone's pieces of stage gs have landed, slot (gs - 1) % NS is free
      PG_STAMP(0, gs, 2);
      if (is_gs < total) issue();                            // stage gs + DIST -> the slot stage gs - 1 has just left
      const unsigned as = smem_base + (unsigned)((gs % NS) * SM::kStage);
      PG_STAMP(0, gs, 3);
      __builtin_amdgcn_s_setprio(1);
      if (!(dbg & 2)) compute(as);
      __builtin_amdgcn_s_setprio(0);
      PG_STAMP(0, gs, 4);
      if (++kt == nk) {
        const int lin = tile_i * G + pos, tm = lin / ntn, tn = lin - tm * ntn;
        stage_out(smem_base + (unsigned)(SM::kRing + (tile_i & 1) * SM::kOut), tn * BN);
        zero_acc();
        kt = 0; ++tile_i;
      }
      PG_STAMP(0, gs, 5);
    }
    RAW_BARRIER();                                           // barrier `total`: the last tile is staged
    return;
  }

  // ================================================================================================ epilogue waves
  const int et = t - 256;
  bf16_t* __restrict__ D = reinterpret_cast<bf16_t*>(p.D);
  const bf16_t* __restrict__ R = reinterpret_cast<const bf16_t*>(p.residual);
  constexpr bool stats = EPI == 1;
  const bool plain = !R && !p.accumulate && !p.relu;         // the staged words go out as they are
  const int Q = (IT + nk - 2) / (nk - 1);                    // chunks per thread and K step: a tile is drained in nk - 1 steps
  const int ec = et % CPR, er0 = et / CPR;                   // this thread's chunk column (fixed: 256 % CPR == 0) and first row
  float sn = 0.f, smean[CH], sm2[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { smean[e] = 0.f; sm2[e] = 0.f; }

  auto tile_coords = [&](int i, int& tm, int& tn) { const int lin = i * G + pos; tm = lin / ntn; tn = lin - tm * ntn; };
  // chunks [k0, k1) of this thread for tile i
  auto drain = [&](int i, int k0, int k1) {
    int tm, tn; tile_coords(i, tm, tn);
    const char* outs = smem + SM::kRing + (i & 1) * SM::kOut;
    const int n = tn * BN + ec * CH;
    for (int k = k0; k < k1; ++k) {
      const int r = er0 + k * (256 / CPR);
      const int m = tm * BM + r;
      if (m >= M || n >= p.Nout) continue;
      const uint4 q = *reinterpret_cast<const uint4*>(outs + out_swz<BN>(r, ec));
      const size_t g = (size_t)m * p.ldd + n;
      float v[CH];
      if (stats || !plain) Chunk<bf16_t>::unpack(q, v);
      if (stats) {
        sn += 1.f; const float inv = 1.f / sn;
#pragma unroll
        for (int e = 0; e < CH; ++e) { const float d = v[e] - smean[e]; smean[e] += d * inv; sm2[e] += d * (v[e] - smean[e]); }
      }
      if (dbg & 1) continue;
      if (plain) { *reinterpret_cast<uint4*>(D + g) = q; continue; }
      if (R) { float w[CH]; Chunk<bf16_t>::load(R + g, w);
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] += w[e]; }
      if (p.accumulate) {
        uint4 qa = *reinterpret_cast<const uint4*>(D + g);   // (mask applied on the packed words: see igemm.hip / DESIGN.md section 7)
        if (p.acc_mask) qa = keep_masked<bf16_t>(qa, p.acc_mask[g / CH]);
        float w[CH]; Chunk<bf16_t>::unpack(qa, w);
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] += w[e]; }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] = v[e] < 0.f ? 0.f : v[e]; }
      Chunk<bf16_t>::store(D + g, v);
    }
  };
  // statistics of tile i: fold this thread's rows with the other row lanes of its wave, leave the wave's record in the tile's
  // own staging tile (every wave has finished reading it: the barrier in front of this step), restart the running sums
  auto stats_scratch = [&](int i) {
    if constexpr (stats) {
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
      float* sp = reinterpret_cast<float*>(smem + SM::kRing + (i & 1) * SM::kOut);      // [4 waves][BN][3]
      static_assert(4 * BN * 3 * 4 <= SM::kOut, "statistics scratch must fit in a staging tile");
      if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          float* q = sp + ((size_t)(wave - 4) * BN + ec * CH + e) * 3;
          q[0] = sn; q[1] = smean[e]; q[2] = sm2[e];
        }
      }
      sn = 0.f;
#pragma unroll
      for (int e = 0; e < CH; ++e) { smean[e] = 0.f; sm2[e] = 0.f; }
      LDS_WAIT_ALL();
    }
  };
  auto stats_fold = [&](int i) {
    if constexpr (stats) {
      int tm, tn; tile_coords(i, tm, tn);
      const float* sp = reinterpret_cast<const float*>(smem + SM::kRing + (i & 1) * SM::kOut);
      if (et < BN && tn * BN + et < p.Nout) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float* q = sp + ((size_t)w * BN + et) * 3;
          const float nb = q[0];
          if (nb > 0.f) { const float nt = n + nb, f = nb / nt, d = q[1] - mean; mean += d * f; m2 += q[2] + d * d * n * f; n = nt; }
        }
        float* out = p.stat_partial + ((size_t)tm * p.Nout + tn * BN + et) * 3;
        out[0] = n; out[1] = mean; out[2] = m2;
      }
    }
  };

  // virtual steps v = 0 .. total + nk: barrier v is shared with the MFMA waves for v <= total (they leave after barrier `total`)
  for (int v = 0; v <= total + nk; ++v) {
    PG_STAMP(1, v, 0);
    RAW_BARRIER();
    PG_STAMP(1, v, 1);
    const int q = v / nk, kt = v - q * nk;
    const int ta = q - 1;                                    // the tile staged before barrier q * nk
    if (ta >= 0 && ta < my_tiles) {
      if (kt < nk - 1) { const int k0 = kt * Q, k1 = k0 + Q < IT ? k0 + Q : IT; if (k0 < IT) drain(ta, k0, k1); }
      else stats_scratch(ta);
    }
    if (kt == 0 && q - 2 >= 0 && q - 2 < my_tiles) stats_fold(q - 2);
    PG_STAMP(1, v, 2);
  }
}

// ------------------------------------------------------------------------------------ host
template <int BM, int BN, int NS, int EPI>
static void launch_pgemm_epi(GatherArgs& a, int grid, hipStream_t st) {
  constexpr int smem = PgSmem<BM, BN, NS>::kBytes;
  auto kern = pgemm_kernel<BM, BN, NS, EPI>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, a);
}
template <int BM, int BN, int NS>
static void launch_pgemm(GatherArgs& a, int ncu, hipStream_t st) {
  const int ntm = cdiv(a.ph[0].M, BM);
  a.ntn = cdiv(a.Nout, BN);
  a.ph[0].ntm = ntm;
  a.ntiles = ntm * a.ntn;
  a.stat_slices = 0;
  if (a.stat_partial) {
    if (!a.residual && !a.accumulate && (size_t)ntm * a.Nout * 3 * sizeof(float) <= a.stat_bytes) a.stat_slices = ntm;
    else a.stat_partial = nullptr;
  }
  constexpr int per_cu_lds = (160 * 1024) / PgSmem<BM, BN, NS>::kBytes;  // blocks that fit a CU's LDS
  constexpr int per_cu = per_cu_lds < 2 ? per_cu_lds : 2;                 // ... and its registers (512-thread blocks)
  static const int cap = getenv("MI355_PG_PER_CU") ? atoi(getenv("MI355_PG_PER_CU")) : 8;
  int grid = ncu * (per_cu < cap ? per_cu : cap);
  if (grid > a.ntiles) grid = a.ntiles;
  if (a.stat_partial) launch_pgemm_epi<BM, BN, NS, 1>(a, grid, st);
  else launch_pgemm_epi<BM, BN, NS, 0>(a, grid, st);
}

// Does the launch described by `a` (filled as for dispatch_gather) fit this kernel?  1x1, unit stride both ways (GEMM rows =
// NHWC pixels in order), bf16, whole 64-channel K steps and at least two of them, no BatchNorm-backward epilogue, no heat-map
// output.
bool pgemm_eligible(const GatherArgs& a, int elem_size) {
  static const int on = getenv("MI355_PGEMM") ? atoi(getenv("MI355_PGEMM")) : 1;
  if (!on || elem_size != 2) return false;
  if (a.nphase != 1 || a.ph[0].ntaps != 1) return false;
  const Tap& tp = a.taps[a.ph[0].tap0];
  if (tp.dy != 0 || tp.dx != 0 || tp.widx != 0) return false;
  if (a.in_sx != 1 || a.in_sy != 1 || a.out_sx != 1 || a.out_sy != 1) return false;
  if (a.ph[0].OHp != a.Hi || a.ph[0].OWp != a.Wi || a.Ho != a.Hi || a.Wo != a.Wi || a.ph[0].out_oy || a.ph[0].out_ox) return false;
  if (a.Ci % 64 || a.Ci < 128 || a.ldb != a.Ci || a.ldd != a.Nout || a.Nout % 8) return false;      // (two K steps at least: kernel schedule)
  if (a.bnb_partial || a.hw) return false;
  const long Mrows = a.ph[0].M;
  if (Mrows * a.Ci * 2 >= (1L << 31) || Mrows * a.Nout * 2 >= (1L << 31)) return false;
  static const int min_rows = getenv("MI355_PGEMM_MIN_ROWS") ? atoi(getenv("MI355_PGEMM_MIN_ROWS")) : 256;
  return Mrows >= min_rows;
}

int dispatch_pgemm(GatherArgs& a, hipStream_t st) {
  static const int dbg = getenv("MI355_PG_DEBUG") ? atoi(getenv("MI355_PG_DEBUG")) : 0;      // diagnostic timing runs: results are wrong
  a.hw = dbg;
  if (dbg & 8) a.bnb_partial = reinterpret_cast<float*>(strtoull(getenv("MI355_PG_DEBUG_PTR") ? getenv("MI355_PG_DEBUG_PTR") : "0", nullptr, 0));
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      MI_FAIL(MI355_ELAUNCH, "pgemm: cannot read the CU count");
    ncu = v;
  }
  const long M = a.ph[0].M;
  static const int force = getenv("MI355_PG_TILE") ? atoi(getenv("MI355_PG_TILE")) : -1;     // experiment switch
  const long t128 = cdiv(M, 128L) * cdiv(a.Nout, 128);
  int sel;
  if (force >= 0) sel = force;
  else if (a.Nout <= 64) sel = cdiv(M, 128L) >= 2 * ncu ? 2 : 3;
  else if (t128 >= 2 * ncu) sel = 0;
  else if (cdiv(M, 64L) * cdiv(a.Nout, 128) >= 2 * ncu) sel = 1;
  else sel = 3;
  switch (sel) {
    case 0: launch_pgemm<128, 128, 3>(a, ncu, st); break;
    case 1: launch_pgemm<64, 128, 3>(a, ncu, st); break;
    case 2: launch_pgemm<128, 64, 3>(a, ncu, st); break;
    case 3: launch_pgemm<64, 64, 3>(a, ncu, st); break;
    case 4: launch_pgemm<128, 128, 2>(a, ncu, st); break;
    case 5: launch_pgemm<64, 128, 2>(a, ncu, st); break;
    case 6: launch_pgemm<64, 64, 2>(a, ncu, st); break;
    case 7: launch_pgemm<64, 128, 4>(a, ncu, st); break;
    case 8: launch_pgemm<64, 64, 4>(a, ncu, st); break;
    default: MI_FAIL(MI355_EINVAL, "pgemm: MI355_PG_TILE=%d", sel);
  }
  MI_CHECK_LAUNCH("pgemm");
  return MI355_OK;
}
