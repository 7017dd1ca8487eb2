// Persistent pipelined GEMM for the 1x1 / unit-stride convolutions (forward and input gradient), bf16, gfx950.
//
//   D[m][n] = epilogue( sum_k A[m][k] * B[n][k] )      A: [M][K] activations (NHWC rows ARE the GEMM rows: no gather),
//                                                      B: [N][K] packed weights, K = channels (multiple of 64)
//
// A second kernel beside gather_gemm_kernel (igemm.hip), OFF by default (MI355_PGEMM / mi355_set_pgemm).  Timed per layer in
// isolation (graph replay, operands warm in L2 / Infinity Cache) it is 15 - 30 % faster on the K-heavy layers with M <= 16 K rows
// and on narrow outputs, and slower where the BatchNorm-statistics epilogue dominates (profiles/r03_pgemm_phases.txt); inside the
// training iteration and the inference forward, where operands arrive from HBM, it gains nothing: 33.31 vs 33.22 / 33.40 ms with
// the selective policy (mode 1), 33.21 vs 32.65 ms and 20.5 k vs 21.5 k images/s when it takes every launch it fits (mode 2).  It
// stays in the tree as the measured answer to "a persistent / weights-stationary kernel for the 1x1 convs", with its parity
// tests (tests/test_gpu_kernels.py: identical bits to the gather kernel), not on the product path.  What it does:
//   * the operands are plain row-major matrices, so a K step's global addresses are ONE per-lane offset per 1-KiB piece, fixed
//     for the whole tile, plus a scalar (k * 128 bytes) in the instruction's soffset: no per-step vector arithmetic;
//   * tiles travel global -> LDS by LDS-DMA (buffer_load ... lds) into a ring of NS stages with NS-1 K steps in flight (counted
//     vmcnt), ONE barrier per K step (a raw s_barrier: the fences of __syncthreads() would drain the ring), no staging
//     registers, no ds_write; a K step's fragment reads and MFMAs are one hand-scheduled asm statement;
//   * the kernel is PERSISTENT: a block walks a list of tiles and the ring never drains -- the first K steps of the next tile are
//     in flight while the current tile's epilogue streams out (staged in the ring slot its last K step has just freed);
//   * same epilogue menu as the gather kernel: bias, device scalar (gradient-layer lambda), residual, accumulate (+ bit mask),
//     ReLU (inference), BatchNorm statistics of the output (EPI = 1: kept for completeness and tests, not dispatched by default).
// What was tried beyond this form and measured slower (wave-specialised MFMA / epilogue / loader roles on a static per-K-step
// barrier schedule) is in profiles/r03_pgemm_phases.txt together with the in-kernel stamps that explain why: with 64 - 96 KB of
// ring per CU the operand fill (~30 B/clk per CU, whichever waves issue it) bounds these layers, not the schedule.
// Fragment layout, LDS row swizzle and MFMA operand order are those of gather_gemm_kernel (128-byte rows, chunk ^ (row >> 1) & 7,
// D^T accumulators); both kernels produce the same bits for the same element.
#include "common.h"
#include <stdlib.h>

#include "igemm_common.h"

template <int BM, int BN, int NS>
struct PgSmem {
  static constexpr int kStage = (BM + BN) * 128;
  static constexpr int kBytes = NS * kStage;
  static_assert(BM * BN * 2 <= kStage, "the epilogue stages the output tile in one ring slot");
};

// swizzled byte offset of 16-byte chunk c of output-tile row r in the epilogue staging image (rows of BN * 2 bytes, no padding):
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

template <int BM, int BN, int NS, int EPI>
__global__ __launch_bounds__(256) void pgemm_kernel(const GatherArgs p) {
  constexpr int CH = 8, NTHR = 256;
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int PA = BM / 32, PB = BN / 32;                  // 1-KiB LDS-DMA pieces per wave and stage
  constexpr int DIST = NS - 1;                               // K steps in flight
  constexpr int CPR = BN / CH;                               // 16-byte chunks per output row
  constexpr int IT = BM * CPR / NTHR;                        // chunks per thread and tile in the epilogue
  constexpr int RSTEP = NTHR / CPR;                          // rows between a thread's consecutive chunks
  static_assert(BM * CPR % NTHR == 0, "whole passes only");
  using SM = PgSmem<BM, BN, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const unsigned smem_base = lds_addr(smem);
  const int lc = t & 7, lr = t >> 3;
  const int lcs = lc ^ ((lr >> 1) & 7);                      // the swizzle is applied on the SOURCE address
  const int r31 = lane & 31, hi = lane >> 5;
  const int M = p.ph[0].M, K = p.Ci, nk = K >> 6, ntn = p.ntn;
  const int G = (int)gridDim.x;
  const int pos = xcd_remap((int)blockIdx.x, G);             // blocks of one XCD take neighbouring tiles (same A rows)
  const int my_tiles = pos < p.ntiles ? (p.ntiles - pos + G - 1) / G : 0;
  const int total = my_tiles * nk;
  if (total == 0) return;

  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  (void)rsA; (void)rsB;                                      // (only used in the device pass)

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
  auto issue = [&]() {                                       // stage is_gs -> ring slot is_gs % NS
    const int slot = is_gs % NS;
    const int soff = is_kt * 128;
    (void)slot; (void)soff; (void)wave_u;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* ldsp;
    char* sa = smem + slot * SM::kStage + wave_u * 1024;
    char* sb = sa + BM * 128;
#pragma unroll
    for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsp)(sa + j * 4096), 16, voffA[j], soff, 0, 0);
#pragma unroll
    for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsp)(sb + j * 4096), 16, voffB[j], soff, 0, 0);
#endif
    ++is_gs;
    if (++is_kt == nk) { is_kt = 0; ++is_tile; if (is_tile < my_tiles) set_issue_tile(is_tile); }
  };

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;       // (defined values for the first asm statement; every tile's first MFMAs take C = 0)
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
#define PG_MFMAZ(ACC, B, A) "v_mfma_f32_32x32x16_bf16 " ACC ", " B ", " A ", 0\n"
#define PG_RD(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n"
  auto compute = [&](unsigned stage_addr, int first) {      // first: the tile's first K step (its first MFMAs take C = 0)
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
          "s_cmp_lg_u32 %28, 0\n\ts_cbranch_scc1 1f\n"
          PG_MFMA("%0", "%6", "%4") PG_MFMA("%1", "%7", "%4") PG_MFMA("%2", "%6", "%5") PG_MFMA("%3", "%7", "%5")
          "s_branch 2f\n1:\n"
          PG_MFMAZ("%0", "%6", "%4") PG_MFMAZ("%1", "%7", "%4") PG_MFMAZ("%2", "%6", "%5") PG_MFMAZ("%3", "%7", "%5")
          "2:\n"
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
            "v"(xb[0][0]), "v"(xb[0][1]), "v"(xb[0][2]), "v"(xb[0][3]), "v"(xb[1][0]), "v"(xb[1][1]), "v"(xb[1][2]), "v"(xb[1][3]), "s"(first)
          : "memory", "scc");
    } else if constexpr (MT == 1 && NT == 2) {
      bf16x8_t a0, b0, b1, c0, d0, d1;
      asm volatile(
          PG_RD("%2", "%8") PG_RD("%3", "%12") PG_RD("%4", "%16")
          PG_RD("%5", "%9") PG_RD("%6", "%13") PG_RD("%7", "%17")
          "s_waitcnt lgkmcnt(3)\n"
          "s_cmp_lg_u32 %20, 0\n\ts_cbranch_scc1 1f\n"
          PG_MFMA("%0", "%3", "%2") PG_MFMA("%1", "%4", "%2")
          "s_branch 2f\n1:\n"
          PG_MFMAZ("%0", "%3", "%2") PG_MFMAZ("%1", "%4", "%2")
          "2:\n"
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
            "v"(xb[0][0]), "v"(xb[0][1]), "v"(xb[0][2]), "v"(xb[0][3]), "v"(xb[1][0]), "v"(xb[1][1]), "v"(xb[1][2]), "v"(xb[1][3]), "s"(first)
          : "memory", "scc");
    } else if constexpr (MT == 2 && NT == 1) {
      bf16x8_t a0, a1, b0, c0, c1, d0;
      asm volatile(
          PG_RD("%2", "%8") PG_RD("%3", "%12") PG_RD("%4", "%16")
          PG_RD("%5", "%9") PG_RD("%6", "%13") PG_RD("%7", "%17")
          "s_waitcnt lgkmcnt(3)\n"
          "s_cmp_lg_u32 %20, 0\n\ts_cbranch_scc1 1f\n"
          PG_MFMA("%0", "%4", "%2") PG_MFMA("%1", "%4", "%3")
          "s_branch 2f\n1:\n"
          PG_MFMAZ("%0", "%4", "%2") PG_MFMAZ("%1", "%4", "%3")
          "2:\n"
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
            "v"(xb[0][0]), "v"(xb[0][1]), "v"(xb[0][2]), "v"(xb[0][3]), "s"(first)
          : "memory", "scc");
    } else {
      bf16x8_t a0, b0, c0, d0;
      asm volatile(
          PG_RD("%1", "%5") PG_RD("%2", "%9") PG_RD("%3", "%6") PG_RD("%4", "%10")
          "s_waitcnt lgkmcnt(2)\n"
          "s_cmp_lg_u32 %13, 0\n\ts_cbranch_scc1 1f\n"
          PG_MFMA("%0", "%2", "%1")
          "s_branch 2f\n1:\n"
          PG_MFMAZ("%0", "%2", "%1")
          "2:\n"
          PG_RD("%1", "%7") PG_RD("%2", "%11")
          "s_waitcnt lgkmcnt(2)\n"
          PG_MFMA("%0", "%4", "%3")
          PG_RD("%3", "%8") PG_RD("%4", "%12")
          "s_waitcnt lgkmcnt(2)\n"
          PG_MFMA("%0", "%2", "%1")
          "s_waitcnt lgkmcnt(0)\n"
          PG_MFMA("%0", "%4", "%3")
          : "+v"(acc[0][0]), "=&v"(a0), "=&v"(b0), "=&v"(c0), "=&v"(d0)
          : "v"(xa[0][0]), "v"(xa[0][1]), "v"(xa[0][2]), "v"(xa[0][3]), "v"(xb[0][0]), "v"(xb[0][1]), "v"(xb[0][2]), "v"(xb[0][3]), "s"(first)
          : "memory", "scc");
    }
  };
#undef PG_MFMA
#undef PG_MFMAZ

  // ---- epilogue of tile (tm, tn): (acc + bias) * scale -> bf16 -> ring slot `outs` -> coalesced 16-byte rows (+ the extras)
  // staging offsets of this lane's 4-wide column runs (row part for i = 0; 32 rows further down the swizzle pattern repeats)
  unsigned so[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) { const int nl = wn0 + j * 32 + 8 * g + 4 * hi; so[j][g] = (unsigned)(out_swz<BN>(wm0 + r31, nl >> 3) + (nl & 7) * 2); }
  const float scale = p.scale ? *p.scale : 1.0f;
  const bool affine = p.bias != nullptr || p.scale != nullptr;
  bf16_t* __restrict__ D = reinterpret_cast<bf16_t*>(p.D);
  const bf16_t* __restrict__ R = reinterpret_cast<const bf16_t*>(p.residual);
  const bool plain = !R && !p.accumulate && !p.relu;         // the staged words go out as they are
  const int ec = t % CPR, er0 = t / CPR;                     // this thread's chunk column (fixed: 256 % CPR == 0) and first row
  const unsigned rd0 = (unsigned)out_swz<BN>(er0, ec);       // chunk k of the thread sits RSTEP rows below chunk k - 1: a constant distance
  auto epilogue = [&](unsigned outs, int tm, int tn) {
    const int m0 = tm * BM, n0 = tn * BN;
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");       // the last MFMAs' results before hipcc's reads of them (asm is opaque to its hazard pass)
    RAW_BARRIER();                                           // every wave has finished reading this slot (last K step)
    if (!affine) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            union { bf16_t h[4]; uint2 q; } u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u.h[e] = (bf16_t)acc[i][j][4 * g + e];
            lds_write8(outs + so[j][g] + (unsigned)(i * 32 * BN * 2), u.q);
          }
    } else {
      float4 bq[NT][4];                                      // bias of this lane's column runs, fetched up front
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn0 + j * 32 + 8 * g + 4 * hi;  // Nout is a multiple of 8: a run is inside or outside as a whole
          bq[j][g] = (p.bias && n < p.Nout) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float bb[4] = {bq[j][g].x, bq[j][g].y, bq[j][g].z, bq[j][g].w};
            union { bf16_t h[4]; uint2 q; } u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u.h[e] = (bf16_t)((acc[i][j][4 * g + e] + bb[e]) * scale);
            lds_write8(outs + so[j][g] + (unsigned)(i * 32 * BN * 2), u.q);
          }
    }
    LDS_WAIT_ALL();
    RAW_BARRIER();
    constexpr bool stats = EPI == 1;
    float sn = 0.f, smean[CH], sm2[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { smean[e] = 0.f; sm2[e] = 0.f; }
    const int n = n0 + ec * CH;
    const int rows_left = M - (m0 + er0);                    // chunk k is inside the matrix iff k * RSTEP < rows_left
    const size_t g0 = (size_t)(m0 + er0) * p.ldd + n;
    u32x4_t q[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) q[k] = lds_read16u(outs + rd0 + (unsigned)(k * RSTEP * BN * 2));
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      // the k-th staged chunk has arrived once only the IT - 1 - k younger reads are outstanding
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(q[k]) : "n"(IT - 1 - k));
      if (k * RSTEP >= rows_left || n >= p.Nout) continue;
      const size_t g = g0 + (size_t)k * RSTEP * p.ldd;
      const uint4 qk = make_uint4(q[k][0], q[k][1], q[k][2], q[k][3]);
      if (!stats && plain) { *reinterpret_cast<uint4*>(D + g) = qk; continue; }
      float v[CH]; Chunk<bf16_t>::unpack(qk, v);
      if (stats) {
        sn += 1.f; const float inv = 1.f / sn;
#pragma unroll
        for (int e = 0; e < CH; ++e) { const float d = v[e] - smean[e]; smean[e] += d * inv; sm2[e] += d * (v[e] - smean[e]); }
        if (plain) { *reinterpret_cast<uint4*>(D + g) = qk; continue; }
      }
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
    if constexpr (stats) {
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
      RAW_BARRIER();                                         // everyone is done reading the staged tile (reads were waited for above)
      static_assert(NW * BN * 3 * 4 <= SM::kStage, "statistics scratch must fit in a ring slot");      // [NW][BN][3] floats
      if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          const unsigned qa = outs + (unsigned)(((wave * BN + ec * CH + e) * 3) * 4);
          lds_write4(qa, sn); lds_write4(qa + 4, smean[e]); lds_write4(qa + 8, sm2[e]);
        }
      }
      LDS_WAIT_ALL();
      RAW_BARRIER();
      if (t < BN && n0 + t < p.Nout) {
        float qn[NW], qm[NW], qv[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const unsigned qa = outs + (unsigned)(((w * BN + t) * 3) * 4);
          qn[w] = lds_read4(qa); qm[w] = lds_read4(qa + 4); qv[w] = lds_read4(qa + 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]), "+v"(qm[0]), "+v"(qm[1]), "+v"(qm[2]),
                     "+v"(qm[3]), "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3]));
        float cn = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const float nb = qn[w];
          if (nb > 0.f) { const float nt = cn + nb, f = nb / nt, d = qm[w] - mean; mean += d * f; m2 += qv[w] + d * d * cn * f; cn = nt; }
        }
        float* out = p.stat_partial + ((size_t)tm * p.Nout + n0 + t) * 3;
        out[0] = cn; out[1] = mean; out[2] = m2;
      }
    }
  };

  // ---- the pipeline
  set_issue_tile(0);
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (is_gs < total) issue();
  int kt = 0, tile_i = 0;
  for (int gs = 0; gs < total; ++gs) {
    // this wave's pieces of stage gs have landed once at most the pieces of the DIST-1 younger stages are outstanding
    // (LDS-DMA, loads and stores retire in issue order; anything the epilogue issued meanwhile is younger still: waiting for
    // more than needed is safe, never for less)
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
    RAW_BARRIER();                                           // everyone's pieces have; slot (gs - 1) % NS is no longer read
    if (is_gs < total) issue();                              // stage gs + DIST -> the slot stage gs - 1 has just left
    const unsigned as = smem_base + (unsigned)((gs % NS) * SM::kStage);
    __builtin_amdgcn_s_setprio(1);
    compute(as, __builtin_amdgcn_readfirstlane(kt == 0 ? 1 : 0));
    __builtin_amdgcn_s_setprio(0);
    if (++kt == nk) {
      const int lin = tile_i * G + pos, tm = lin / ntn, tn = lin - tm * ntn;
      epilogue(as, tm, tn);
      kt = 0; ++tile_i;
    }
  }
}

// ------------------------------------------------------------------------------------ host
template <int BM, int BN, int NS, int EPI>
static void launch_pgemm_epi(GatherArgs& a, int grid, hipStream_t st) {
  constexpr int smem = PgSmem<BM, BN, NS>::kBytes;
  auto kern = pgemm_kernel<BM, BN, NS, EPI>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, a);
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
  constexpr int per_cu = per_cu_lds < 4 ? per_cu_lds : 4;
  static const int cap = getenv("MI355_PG_PER_CU") ? atoi(getenv("MI355_PG_PER_CU")) : 8;
  int grid = ncu * (per_cu < cap ? per_cu : cap);
  if (grid > a.ntiles) grid = a.ntiles;
  if (a.stat_partial) launch_pgemm_epi<BM, BN, NS, 1>(a, grid, st);
  else launch_pgemm_epi<BM, BN, NS, 0>(a, grid, st);
}

// Does the launch described by `a` (filled as for dispatch_gather) fit this kernel?  1x1, unit stride both ways (GEMM rows =
// NHWC pixels in order), bf16, whole 64-channel K steps, no BatchNorm-backward epilogue, no heat-map output.
static int g_pgemm_mode = -1;        // run-time switch (mi355_set_pgemm); -1: the environment decides (MI355_PGEMM, default 0)
// 0: never; 1: where it measured faster than the gather kernel (below); 2: wherever the launch fits the kernel (tests, A/B runs).
// Returns the previous setting.
extern "C" int mi355_set_pgemm(int mode) {
  const int prev = g_pgemm_mode;
  g_pgemm_mode = mode < 0 ? -1 : (mode > 2 ? 2 : mode);
  return prev;
}
bool pgemm_eligible(const GatherArgs& a, int elem_size) {
  static const int env_mode = getenv("MI355_PGEMM") ? atoi(getenv("MI355_PGEMM")) : 0;
  const int mode = g_pgemm_mode >= 0 ? g_pgemm_mode : env_mode;
  if (!mode || elem_size != 2) return false;
  if (a.nphase != 1 || a.ph[0].ntaps != 1) return false;
  const Tap& tp = a.taps[a.ph[0].tap0];
  if (tp.dy != 0 || tp.dx != 0 || tp.widx != 0) return false;
  if (a.in_sx != 1 || a.in_sy != 1 || a.out_sx != 1 || a.out_sy != 1) return false;
  if (a.ph[0].OHp != a.Hi || a.ph[0].OWp != a.Wi || a.Ho != a.Hi || a.Wo != a.Wi || a.ph[0].out_oy || a.ph[0].out_ox) return false;
  if (a.Ci % 64 || a.ldb != a.Ci || a.ldd != a.Nout || a.Nout % 8) return false;
  if (a.bnb_partial || a.hw) return false;
  const long Mrows = a.ph[0].M;
  if (Mrows * a.Ci * 2 >= (1L << 31) || Mrows * a.Nout * 2 >= (1L << 31)) return false;
  if (Mrows < 256) return false;
  if (mode >= 2) return true;
  // Where it wins (profiles/r03_pgemm_phases.txt, graph-timed per layer): not with the BatchNorm-statistics epilogue (one or two
  // blocks per CU cannot hide its VALU work behind other blocks' MFMAs as the gather kernel's three or four do); otherwise on
  // K-heavy layers, narrow outputs and small row counts; K = 64 .. 256 with wide outputs and many rows is HBM-bound on both
  // kernels and the gather kernel's higher occupancy keeps more bytes in flight there.
  if (a.stat_partial) return false;
  return a.Ci >= 512 || a.Nout <= 64 || Mrows <= 16384;
}

int dispatch_pgemm(GatherArgs& a, hipStream_t st) {
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
