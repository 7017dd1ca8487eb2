// Weights-stationary streaming GEMM for the 1x1 / unit-stride convolutions (forward and input gradient), bf16, gfx950.
//
//   D[m][n] = epilogue( sum_k A[m][k] * B[n][k] )      A: [M][K] activations (NHWC rows ARE the GEMM rows: no gather),
//                                                      B: [N][K] packed weights, K = channels (multiple of 64)
//
// The 1x1 convs at 64x64 / 32x32 maps are HBM-bound (a few hundred FLOP per byte at most), yet the gather kernel runs them at
// 2.5 TB/s: with 64x128 / 128x128 output tiles every tile re-stages its slice of the weights, so two thirds of the bytes that
// enter a CU's LDS are weights (256 -> 256 @64x64, B = 64: 786 MB through the CUs' vector-memory path for 268 MB of HBM
// traffic), and that path takes in ~30 B/clk per CU whichever waves issue the loads (profiles/r03_pgemm_phases.txt) -- the
// launch is bound by LDS fill, not by HBM.  Round 3's persistent ring kernel streamed both operands through its ring and hit
// the same wall.  This kernel keeps the block's WEIGHT SLICE RESIDENT in LDS instead:
//   * a persistent block owns ONE column tile (BN output channels) for its whole life: its [BN][K] weight slice is fetched once
//     (K * BN * 2 bytes <= 64 KB) and stays; the block then walks the row tiles gm, gm + Gm, ... of that column tile;
//   * only activations stream: an LDS-DMA ring of NS stages of BM x 64 channels (buffer_load ... lds, swizzle applied on the
//     source address), NS - 1 K steps in flight across tile boundaries (counted vmcnt), ONE raw s_barrier per K step; the bytes
//     entering LDS per output tile drop to the activation tile itself = the HBM bytes;
//   * the epilogue stages the tile in a region of its own (the ring never drains: the next tiles' activations are in flight
//     while a tile's epilogue streams out); same epilogue menu as the gather kernel -- bias, device scalar (gradient-layer
//     lambda), residual, accumulate (+ bit mask), ReLU (inference), BatchNorm statistics of the output (EPI = 1);
//   * a K step's fragment reads and MFMAs are one hand-scheduled asm statement (hipcc sank the MFMAs below the waits and
//     shuttled the accumulators between the register files around the conditional epilogue).
// Fragment layout, LDS row swizzle and MFMA operand order are those of gather_gemm_kernel (128-byte rows, chunk ^ (row >> 1) & 7,
// D^T accumulators); both kernels produce the same bits for the same element (tests/test_gpu_kernels.py).
#include "common.h"
#include <stdlib.h>

#include "igemm_common.h"

// LDS of a block: [weight slice: nk K steps x BN rows x 128 B][ring: NS stages x BM rows x 128 B][output staging: BM x BN bf16]
template <int BM, int BN, int NS>
struct PgSmem {
  static constexpr int kStage = BM * 128;
  static constexpr int kOut = BM * BN * 2;
  static constexpr int bytes(int nk) { return nk * BN * 128 + NS * kStage + kOut; }
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

template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N > 63 ? 63 : N) : "memory");
}
// s_waitcnt vmcnt(a) for a wave-uniform run-time allowance: the immediate is the largest even number <= min(a, 62)
__device__ __forceinline__ void wait_vm_dyn(int a) {
  switch (a >> 1) {
    case 0: wait_vm<0>(); break;   case 1: wait_vm<2>(); break;   case 2: wait_vm<4>(); break;   case 3: wait_vm<6>(); break;
    case 4: wait_vm<8>(); break;   case 5: wait_vm<10>(); break;  case 6: wait_vm<12>(); break;  case 7: wait_vm<14>(); break;
    case 8: wait_vm<16>(); break;  case 9: wait_vm<18>(); break;  case 10: wait_vm<20>(); break; case 11: wait_vm<22>(); break;
    case 12: wait_vm<24>(); break; case 13: wait_vm<26>(); break; case 14: wait_vm<28>(); break; case 15: wait_vm<30>(); break;
    case 16: wait_vm<32>(); break; case 17: wait_vm<34>(); break; case 18: wait_vm<36>(); break; case 19: wait_vm<38>(); break;
    case 20: wait_vm<40>(); break; case 21: wait_vm<42>(); break; case 22: wait_vm<44>(); break; case 23: wait_vm<46>(); break;
    case 24: wait_vm<48>(); break; case 25: wait_vm<50>(); break; case 26: wait_vm<52>(); break; case 27: wait_vm<54>(); break;
    case 28: wait_vm<56>(); break; case 29: wait_vm<58>(); break; case 30: wait_vm<60>(); break; default: wait_vm<62>(); break;
  }
}

// ADD: the build whose epilogue adds a tensor read from global memory (residual / the gradient accumulated onto, with its bit mask).
// A wave consumes its vector-memory results in issue order, so an epilogue load could only be used once every ring stage issued
// before it had landed (one memory latency per tile), and hipcc guards such a load's destination registers with vmcnt(0) in the
// K loop even when the branch is not taken.  So the addends never travel through registers: a tile's addend tile (and mask bytes)
// is fetched by LDS-DMA into a slot of an ADDEND RING, issued IN FRONT OF the activation pieces of the tile's first K step -- older
// than they are, hence landed whenever that stage's counted wait returns -- NS - 1 K steps before the tile's first MFMA; the epilogue
// reads it from LDS.  Slots: floor((NS - 1) / nk) + 2 (GatherArgs.pg_nadd), so a slot is rewritten only after the tile that used
// it has left its epilogue behind a K-step barrier.  A build of its own: the plain builds contain no such traffic at all.
template <int BM, int BN, int NS, int EPI, bool ADD>
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
  const int M = p.ph[0].M, K = p.Ci, nk = K >> 6, ntn = p.ntn, ntm = p.ph[0].ntm;
  const int G = (int)gridDim.x, Gm = G / ntn;                // (host: G is a multiple of ntn)
  const int pos = xcd_remap((int)blockIdx.x, G);             // blocks of one XCD take neighbouring (row group, column tile) pairs: same A rows
  const int tn = pos % ntn, gm = pos / ntn;                  // this block's column tile, and its first row tile
  const int my_tiles = gm < ntm ? (ntm - gm + Gm - 1) / Gm : 0;
  const int total = my_tiles * nk;
  if (total == 0) return;
  const unsigned ring_base = smem_base + (unsigned)(nk * BN * 128), out_base = ring_base + (unsigned)(NS * SM::kStage);
  constexpr int kMask = BM * BN / 8;                         // mask bytes of a tile (one bit per element)
  const int nadd = ADD ? p.pg_nadd : 0;
  const unsigned add_base = out_base + (unsigned)SM::kOut, mask_base = add_base + (unsigned)(nadd * SM::kOut);
  const int add_off = nk * BN * 128 + NS * SM::kStage + SM::kOut;      // (byte offsets of the same two regions from smem)
  const int mask_off = add_off + nadd * SM::kOut;

  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
  (void)rsA; (void)rsB;                                      // (only used in the device pass)

  // ---- issue side: tile / K step of the next LDS-DMA stage to launch
  int is_tile = 0, is_kt = 0, is_gs = 0, is_slot = 0, is_add = 0;
  constexpr int RPI = 1024 / (BN * 2), PADD = BM / (4 * RPI);    // rows of an addend tile per 1-KiB piece; pieces per wave
  constexpr int NMD = kMask / 4;                             // mask dwords of a tile (waves 0 .. NMD / 64 - 1 fetch 64 each)
  const bf16_t* __restrict__ ADDSRC = !ADD ? nullptr : (p.residual ? reinterpret_cast<const bf16_t*>(p.residual)
                                                                  : (p.accumulate ? reinterpret_cast<const bf16_t*>(p.D) : nullptr));
  const bool use_mask = ADD && p.accumulate && !p.residual && p.acc_mask != nullptr;
  const __amdgpu_buffer_rsrc_t rsADD = make_rsrc(ADD ? (const void*)ADDSRC : p.A, ADD && ADDSRC ? (unsigned)((long)M * p.ldd * 2) : 0u);
  const __amdgpu_buffer_rsrc_t rsMSK = make_rsrc(use_mask ? (const void*)p.acc_mask : p.A, use_mask ? (unsigned)((long)M * p.ldd / 8) : 0u);
  (void)rsADD; (void)rsMSK;
  int voffA[PA];
  auto set_issue_tile = [&](int i) {
    const int tm = gm + i * Gm;
#pragma unroll
    for (int j = 0; j < PA; ++j) { const int m = tm * BM + j * 32 + lr; voffA[j] = m < M ? (m * K + lcs * CH) * 2 : OOB_OFF; }
  };
  auto issue = [&]() {                                       // stage is_gs -> ring slot is_gs % NS
    const int soff = is_kt * 128;
    (void)soff; (void)wave_u;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* ldsp;
    if constexpr (ADD) {
      if (is_kt == 0 && ADDSRC) {        // (block-uniform) this tile's addend tile and mask bytes first: older than the stage's pieces
        const int tm = gm + is_tile * Gm;
        char* sd = smem + add_off + is_add * SM::kOut;
#pragma unroll
        for (int j = 0; j < PADD; ++j) {
          const int q = j * 4 + wave_u;                      // piece q = rows q * RPI .. of the tile, row-major image
          const int m = tm * BM + q * RPI + lane / CPR, n = tn * BN + (lane % CPR) * CH;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsADD, (ldsp)(sd + q * 1024), 16, (m < M && n < p.Nout) ? (m * p.ldd + n) * 2 : OOB_OFF, 0, 0, 0);
        }
        if (use_mask && wave_u * 64 < NMD) {
          const int d = wave_u * 64 + lane;                  // dword d of the tile's [BM][BN / 8] mask bytes
          const int m = tm * BM + d / (BN / 32), nb = tn * (BN / 8) + (d % (BN / 32)) * 4;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsMSK, (ldsp)(smem + mask_off + is_add * kMask + wave_u * 256), 4,
                                                   (m < M && nb * 8 < p.Nout) ? m * (p.ldd / 8) + nb : OOB_OFF, 0, 0, 0);
        }
        if (++is_add == nadd) is_add = 0;
      }
    }
    char* sa = smem + nk * BN * 128 + is_slot * SM::kStage + wave_u * 1024;
#pragma unroll
    for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsp)(sa + j * 4096), 16, voffA[j], soff, 0, 0);
#endif
    ++is_gs;
    if (++is_slot == NS) is_slot = 0;
    if (++is_kt == nk) { is_kt = 0; ++is_tile; if (is_tile < my_tiles) set_issue_tile(is_tile); }
  };
  // the block's weight slice, once: K step kt of rows tn * BN .. + BN -> [kt][BN][128 B], same swizzled image as a ring stage
  {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* ldsp;
    int voffB[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) { const int n = tn * BN + j * 32 + lr; voffB[j] = n < p.Nout ? (n * p.ldb + lcs * CH) * 2 : OOB_OFF; }
    for (int kt = 0; kt < nk; ++kt) {
      char* sb = smem + kt * BN * 128 + wave_u * 1024;
#pragma unroll
      for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsp)(sb + j * 4096), 16, voffB[j], kt * 128, 0, 0);
    }
#endif
  }

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
    for (int j = 0; j < NT; ++j) fb[j][s] = (unsigned)swz128(wn0 + j * 32 + r31, 2 * s + hi);
  }
  // One K step = ONE asm statement: 4 x (MT + NT) fragment reads, software-pipelined one 16-deep sub-step ahead of the 4 x MT x NT
  // MFMAs (two fragment buffers), waits counted in lgkmcnt.  hipcc would sink the MFMAs below the later waits and shuttle the
  // accumulators between the two register files around the conditional epilogue; here they stay in the accumulator file.
#define PG_MFMA(ACC, B, A) "v_mfma_f32_32x32x16_bf16 " ACC ", " B ", " A ", " ACC "\n"
#define PG_MFMAZ(ACC, B, A) "v_mfma_f32_32x32x16_bf16 " ACC ", " B ", " A ", 0\n"
#define PG_RD(DST, ADDR) "ds_read_b128 " DST ", " ADDR "\n"
  auto compute = [&](unsigned stage_addr, unsigned w_addr, int first) {      // first: the tile's first K step (its first MFMAs take C = 0)
    unsigned xa[MT][4], xb[NT][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < MT; ++i) xa[i][s] = stage_addr + fa[i][s];
#pragma unroll
      for (int j = 0; j < NT; ++j) xb[j][s] = w_addr + fb[j][s];
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
  const bool plain = (!ADD || (!R && !p.accumulate)) && !p.relu;         // the staged words go out as they are
  const int ec = t % CPR, er0 = t / CPR;                     // this thread's chunk column (fixed: 256 % CPR == 0) and first row
  const unsigned rd0 = (unsigned)out_swz<BN>(er0, ec);       // chunk k of the thread sits RSTEP rows below chunk k - 1: a constant distance
  // bias of this lane's column runs: the block's column tile never changes, so it is fetched once, before the ring starts (a load
  // inside the epilogue could only be used after every ring stage in flight had landed)
  float4 bq[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = tn * BN + wn0 + j * 32 + 8 * g + 4 * hi;  // Nout is a multiple of 8: a run is inside or outside as a whole
      bq[j][g] = (p.bias && n < p.Nout) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  // BatchNorm statistics (EPI = 1): a block owns its column tile for life, so every thread keeps RUNNING sums over all the rows it
  // streams out -- count, and per channel the sum and the sum of squares of the deviations from the first value it saw (3 VALU per
  // element; the deviations keep the sums small, so M2 = q - s^2 / n does not cancel) -- and the block folds them ONCE, after its
  // last tile: one (n, mean, M2) record per block and channel (slice = the block's row group) instead of one per tile.
  constexpr bool stats = EPI == 1;
  float st_n = 0.f, st_ref[CH], st_s[CH], st_q[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { st_ref[e] = 0.f; st_s[e] = 0.f; st_q[e] = 0.f; }
  int cons_add = 0;                                          // slot of the addend ring the next epilogue reads
  auto epilogue = [&](unsigned outs, int tm) {
    const int m0 = tm * BM, n0 = tn * BN;
    const int n = n0 + ec * CH;
    const int rows_left = M - (m0 + er0);                    // chunk k is inside the matrix iff k * RSTEP < rows_left
    const size_t g0 = (size_t)(m0 + er0) * p.ldd + n;
    // addends (residual / the gradient accumulated onto) and mask bytes of this thread's chunks, from the tile's slot of the
    // addend ring (landed since the counted wait of the tile's first K step; published by that step's barrier)
    u32x4_t qadd[ADD ? IT : 1]; unsigned mk[ADD ? IT : 1];
    if constexpr (ADD) if (ADDSRC) {
      const unsigned ab = add_base + (unsigned)(cons_add * SM::kOut) + (unsigned)(er0 * BN * 2 + ec * 16);
      const unsigned mb = mask_base + (unsigned)(cons_add * kMask) + (unsigned)(er0 * (BN / 8) + ec);
#pragma unroll
      for (int k = 0; k < IT; ++k) {
        qadd[k] = lds_read16u(ab + (unsigned)(k * RSTEP * BN * 2));
        mk[k] = 0xffu;
        if (use_mask) { unsigned v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(mb + (unsigned)(k * RSTEP * (BN / 8)))); mk[k] = v; }
      }
      LDS_WAIT_ALL();
      if (++cons_add == nadd) cons_add = 0;
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");       // the last MFMAs' results before hipcc's reads of them (asm is opaque to its hazard pass)
    // (the staging region is this kernel's own: every wave left the previous tile's epilogue reads behind at one of the K-step
    //  barriers since -- at least one per tile)
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
        if (st_n == 0.f) {
#pragma unroll
          for (int e = 0; e < CH; ++e) st_ref[e] = v[e];
        }
        st_n += 1.f;
#pragma unroll
        for (int e = 0; e < CH; ++e) { const float d = v[e] - st_ref[e]; st_s[e] += d; st_q[e] += d * d; }
        if (plain) { *reinterpret_cast<uint4*>(D + g) = qk; continue; }
      }
      if constexpr (ADD) {
        if (ADDSRC) {      // residual, or the gradient accumulated onto (not both: the host sends such a launch to the gather kernel)
          uint4 qa = make_uint4(qadd[k][0], qadd[k][1], qadd[k][2], qadd[k][3]);
          // the bit mask is applied on the packed words (see igemm.hip / DESIGN.md section 7)
          if (use_mask) qa = keep_masked<bf16_t>(qa, mk[k]);
          float w[CH]; Chunk<bf16_t>::unpack(qa, w);
#pragma unroll
          for (int e = 0; e < CH; ++e) v[e] += w[e]; }
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] = v[e] < 0.f ? 0.f : v[e]; }
      Chunk<bf16_t>::store(D + g, v);
    }
  };

  // ---- the pipeline
  set_issue_tile(0);
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (is_gs < total) issue();
  int kt = 0, tile_i = 0, cur_slot = 0;
  unsigned epi_hist = 0;                                      // bit d: the K step d + 1 steps ago ended with a full tile's epilogue
  const bool full_cols = (tn + 1) * BN <= p.Nout;
  for (int gs = 0; gs < total; ++gs) {
    // this wave's pieces of stage gs have landed once at most the pieces of the DIST-1 younger stages are outstanding
    // (LDS-DMA, loads and stores retire in issue order; anything the epilogue issued meanwhile is younger still: waiting for
    // more than needed is safe, never for less)
#if defined(__HIP_DEVICE_COMPILE__)
    if (gs + DIST - 1 < total && DIST > 1) {
      // vmcnt counts this wave's vector-memory instructions of every kind, in order: behind stage gs's pieces came the pieces of
      // the DIST - 1 younger stages AND the output stores of every epilogue run since (IT store instructions per FULL tile --
      // a lower bound: partial tiles, residual / accumulate loads and statistics stores are not counted, which only waits longer).
      // Counting the ring loads alone let a tile's stores eat the allowance: with one K step per tile the ring ran one deep.
      constexpr int W = (DIST - 1) * PA;
      const int c = __builtin_amdgcn_readfirstlane(__builtin_popcount(epi_hist & ((1u << DIST) - 1u)));       // full-tile epilogues within the last DIST steps
      if constexpr (ADD) {
        // ... and the addend pieces issued in front of every younger stage that starts a tile (stage gs + j starts one iff
        // (kt + j) % nk == 0; all DIST - 1 younger stages exist on this branch).  The mask piece is not counted (not every wave has one).
        int starts = 0, kk = kt;
#pragma unroll
        for (int j = 1; j < DIST; ++j) { if (++kk == nk) kk = 0; starts += (kk == 0) ? 1 : 0; }
        wait_vm_dyn(__builtin_amdgcn_readfirstlane(W + c * IT + (ADDSRC ? starts * PADD : 0)));
      } else {
        switch (c) {
          case 0: wait_vm<W>(); break;
          case 1: wait_vm<W + IT>(); break;
          case 2: wait_vm<W + 2 * IT>(); break;
          case 3: wait_vm<W + 3 * IT>(); break;
          case 4: wait_vm<W + 4 * IT>(); break;
          case 5: wait_vm<W + 5 * IT>(); break;
          case 6: wait_vm<W + 6 * IT>(); break;
          default: wait_vm<W + 7 * IT>(); break;
        }
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    RAW_BARRIER();                                           // everyone's pieces have; slot (gs - 1) % NS is no longer read
    if (is_gs < total) issue();                              // stage gs + DIST -> the slot stage gs - 1 has just left
    const unsigned as = ring_base + (unsigned)(cur_slot * SM::kStage);
    if (++cur_slot == NS) cur_slot = 0;
    __builtin_amdgcn_s_setprio(1);
    compute(as, smem_base + (unsigned)(kt * BN * 128), __builtin_amdgcn_readfirstlane(kt == 0 ? 1 : 0));
    __builtin_amdgcn_s_setprio(0);
    epi_hist <<= 1;
    if (++kt == nk) {
      const int tm = gm + tile_i * Gm;
      epilogue(out_base, tm);
      if (full_cols && (tm + 1) * BM <= M) epi_hist |= 1u;
      kt = 0; ++tile_i;
    }
  }
  if constexpr (stats) {
    // ---- the block's statistics record: per thread (n, ref, s, q) -> (n, mean, M2); row lanes of a wave by shuffle-down (lower lane =
    // left operand), the waves in wave order through LDS: a fixed order, bitwise reproducible
    constexpr int NW = NTHR / 64;
    const int n0 = tn * BN;
    float sn = st_n, smean[CH], sm2[CH];
    const float inv = sn > 0.f ? 1.f / sn : 0.f;
#pragma unroll
    for (int e = 0; e < CH; ++e) { smean[e] = st_ref[e] + st_s[e] * inv; sm2[e] = st_q[e] - st_s[e] * st_s[e] * inv; sm2[e] = sm2[e] < 0.f ? 0.f : sm2[e]; }
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RAW_BARRIER();                                           // every wave has left the last epilogue: the staging region is free
    static_assert(NW * BN * 3 * 4 <= SM::kOut, "statistics scratch must fit in the staging region");      // [NW][BN][3] floats
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const unsigned qa = out_base + (unsigned)(((wave * BN + ec * CH + e) * 3) * 4);
        lds_write4(qa, sn); lds_write4(qa + 4, smean[e]); lds_write4(qa + 8, sm2[e]);
      }
    }
    LDS_WAIT_ALL();
    RAW_BARRIER();
    if (t < BN && n0 + t < p.Nout) {
      float qn[NW], qm[NW], qv[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const unsigned qa = out_base + (unsigned)(((w * BN + t) * 3) * 4);
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
      float* out = p.stat_partial + ((size_t)gm * p.Nout + n0 + t) * 3;
      out[0] = cn; out[1] = mean; out[2] = m2;
    }
  }
}

// ------------------------------------------------------------------------------------ host
template <int BM, int BN, int NS, int EPI, bool ADD>
static void launch_pgemm_epi(GatherArgs& a, int ncu, int ntm, int smem, hipStream_t st) {
  auto kern = pgemm_kernel<BM, BN, NS, EPI, ADD>;
  static int attr = 0;       // dynamic-LDS cap raised so far for this instantiation
  if (smem > attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = smem; }
  // blocks that really share a CU at this LDS size (the persistent grid must not exceed what is resident, or late blocks queue)
  static int occ_smem[4] = {0, 0, 0, 0}, occ_n[4] = {0, 0, 0, 0};
  int per_cu = 0;
  for (int i = 0; i < 4; ++i) if (occ_smem[i] == smem) per_cu = occ_n[i];
  if (!per_cu) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, (size_t)smem) != hipSuccess || nb < 1) nb = 1;
    per_cu = nb;
    for (int i = 0; i < 4; ++i) if (!occ_smem[i]) { occ_smem[i] = smem; occ_n[i] = nb; break; }
  }
  static const int cap = getenv("MI355_PG_PER_CU") ? atoi(getenv("MI355_PG_PER_CU")) : 2;
  if (per_cu > cap) per_cu = cap;
  // every block owns one column tile: the grid is a whole number of (row group, column tile) pairs, at most one per row tile
  int groups = ncu * per_cu / a.ntn; if (groups < 1) groups = 1;
  if (groups > ntm) groups = ntm;
  if (EPI == 1) a.stat_slices = groups;      // one statistics record per block and channel (slice = row group; groups <= ntm fits, checked by the caller)
  hipLaunchKernelGGL(kern, dim3(groups * a.ntn), dim3(256), smem, st, a);
}
template <int BM, int BN, int NS>
static bool launch_pgemm(GatherArgs& a, int ncu, hipStream_t st) {
  const int nk = a.Ci >> 6;
  const bool add = a.residual || a.accumulate;
  a.pg_nadd = add ? (NS - 1) / nk + 2 : 0;                    // addend-ring slots (kernel comment): tile + mask bytes each
  const int smem = PgSmem<BM, BN, NS>::bytes(nk) + a.pg_nadd * (PgSmem<BM, BN, NS>::kOut + BM * BN / 8);
  if (smem > 160 * 1024) return false;
  const int ntm = cdiv(a.ph[0].M, BM);
  a.ntn = cdiv(a.Nout, BN);
  a.ph[0].ntm = ntm;
  a.ntiles = ntm * a.ntn;
  a.stat_slices = 0;
  if (a.stat_partial && (a.residual || a.accumulate || (size_t)ntm * a.Nout * 3 * sizeof(float) > a.stat_bytes)) a.stat_partial = nullptr;
  if (a.stat_partial) launch_pgemm_epi<BM, BN, NS, 1, false>(a, ncu, ntm, smem, st);
  else if (a.residual || a.accumulate) launch_pgemm_epi<BM, BN, NS, 0, true>(a, ncu, ntm, smem, st);
  else launch_pgemm_epi<BM, BN, NS, 0, false>(a, ncu, ntm, smem, st);
  return true;
}

// Does the launch described by `a` (filled as for dispatch_gather) fit this kernel?  1x1, unit stride both ways (GEMM rows =
// NHWC pixels in order), bf16, whole 64-channel K steps, a weight slice of one column tile that fits LDS beside the ring, no
// BatchNorm-backward epilogue, no heat-map output, no concatenated second operand.
static int g_pgemm_mode = -1;        // run-time switch (mi355_set_pgemm); -1: the environment decides (MI355_PGEMM, default 1)
// 0: never; 1: where it measured faster than the gather kernel (below); 2: wherever the launch fits the kernel (tests, A/B runs).
// Returns the previous setting.
extern "C" int mi355_set_pgemm(int mode) {
  const int prev = g_pgemm_mode;
  g_pgemm_mode = mode < 0 ? -1 : (mode > 2 ? 2 : mode);
  return prev;
}
static int pg_bn(const GatherArgs& a) {       // column tile: the widest whose weight slice [BN][K] stays within 64 KB
  if (a.Nout > 64 && (long)a.Ci * 128 * 2 <= 64 * 1024) return 128;
  if ((long)a.Ci * 64 * 2 <= 64 * 1024) return 64;
  return 0;
}
bool pgemm_eligible(const GatherArgs& a, int elem_size) {
  static const int env_mode = getenv("MI355_PGEMM") ? atoi(getenv("MI355_PGEMM")) : 1;
  const int mode = g_pgemm_mode >= 0 ? g_pgemm_mode : env_mode;
  if (!mode || elem_size != 2) return false;
  if (a.nphase != 1 || a.ph[0].ntaps != 1 || a.A2) return false;
  const Tap& tp = a.taps[a.ph[0].tap0];
  if (tp.dy != 0 || tp.dx != 0 || tp.widx != 0) return false;
  if (a.in_sx != 1 || a.in_sy != 1 || a.out_sx != 1 || a.out_sy != 1) return false;
  if (a.ph[0].OHp != a.Hi || a.ph[0].OWp != a.Wi || a.Ho != a.Hi || a.Wo != a.Wi || a.ph[0].out_oy || a.ph[0].out_ox) return false;
  if (a.Ci % 64 || a.ldb != a.Ci || a.ldd != a.Nout || a.Nout % 8) return false;
  if (a.bnb_partial || a.hw) return false;
  const long Mrows = a.ph[0].M;
  if (Mrows * a.Ci * 2 >= (1L << 31) || Mrows * a.Nout * 2 >= (1L << 31)) return false;
  if (Mrows < 256 || !pg_bn(a)) return false;
  if (a.residual && a.accumulate) return false;               // (one addend tensor per tile in the addend ring)
  if (a.acc_mask && (a.Nout % 32 || !a.accumulate)) return false;      // mask bytes travel as dwords
  if (mode >= 2) return true;
  // residual / accumulate epilogues (addend ring): one block per CU carries ring, addend slots and a heavier epilogue -- measured
  // 87 -> 74 us on 64 -> 256 @64x64 + residual (eval), 44 -> 46 us on 128 -> 512 @32x32 + residual: worth it for one K step per tile only
  static const int add_maxk = getenv("MI355_PG_ADD") ? atoi(getenv("MI355_PG_ADD")) * 64 : 64;      // A/B switch: largest K taken (0 = none)
  if ((a.residual || a.accumulate) && a.Ci > add_maxk) return false;
  // Where it wins against the gather kernel (profiles/r04_pgemm_layers.txt: per layer, operands from HBM): the large maps
  // (>= 32 K rows), with a short K (<= 128: wide outputs stream at 5 TB/s) or a narrow output (K = 256 -> 64 / 128 columns).
  // K = 256 -> 256 columns needs two column tiles, i.e. the activations twice (no gain), K = 512 leaves room for 64-wide
  // column tiles only (slower).
  static const long min_rows = getenv("MI355_PG_MIN_ROWS") ? atol(getenv("MI355_PG_MIN_ROWS")) : 32768;
  return Mrows >= min_rows && (a.Ci <= 128 || (a.Ci <= 256 && a.Nout <= 128));
}

int dispatch_pgemm(GatherArgs& a, hipStream_t st) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      MI_FAIL(MI355_ELAUNCH, "pgemm: cannot read the CU count");
    ncu = v;
  }
  static const int ring_env = getenv("MI355_PG_RING") ? atoi(getenv("MI355_PG_RING")) : 0;        // experiment switches
  static const int bm = getenv("MI355_PG_BM") ? atoi(getenv("MI355_PG_BM")) : 64;
  const int bn = pg_bn(a);
  const int nk = a.Ci >> 6;
  // ring depth: the deepest ring with which TWO blocks still share a CU (each hides the other's epilogue), else the deepest that fits
  int ring = ring_env;
  if (!ring) {
    const bool add = a.residual || a.accumulate;
    auto bytes = [&](int ns) { return nk * bn * 128 + ns * 64 * 128 + 64 * bn * 2 + (add ? ((ns - 1) / nk + 2) * (64 * bn * 2 + 64 * bn / 8) : 0); };
    if (!add) for (int ns : {8, 6, 5, 4}) if (!ring && bytes(ns) <= 80 * 1024) ring = ns;      // two blocks per CU
    for (int ns : {8, 6, 5, 4}) if (!ring && bytes(ns) <= 160 * 1024) ring = ns;
    if (!ring) ring = 4;
  }
  bool ok = false;
  if (bn == 128) {
    if (bm == 128) ok = launch_pgemm<128, 128, 4>(a, ncu, st);
    else if (ring == 8) ok = launch_pgemm<64, 128, 8>(a, ncu, st);
    else if (ring == 6) ok = launch_pgemm<64, 128, 6>(a, ncu, st);
    else if (ring == 5) ok = launch_pgemm<64, 128, 5>(a, ncu, st);
    else ok = launch_pgemm<64, 128, 4>(a, ncu, st);
  } else {
    if (bm == 128) ok = launch_pgemm<128, 64, 4>(a, ncu, st);
    else if (ring == 8) ok = launch_pgemm<64, 64, 8>(a, ncu, st);
    else if (ring == 6) ok = launch_pgemm<64, 64, 6>(a, ncu, st);
    else if (ring == 5) ok = launch_pgemm<64, 64, 5>(a, ncu, st);
    else ok = launch_pgemm<64, 64, 4>(a, ncu, st);
  }
  if (!ok) MI_FAIL(MI355_EINVAL, "pgemm: the weight slice of K = %d channels does not fit LDS (bn %d)", a.Ci, bn);
  MI_CHECK_LAUNCH("pgemm");
  return MI355_OK;
}
