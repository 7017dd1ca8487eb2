// Argument block and device helpers shared by the implicit-GEMM gather kernels (igemm.hip: bf16 / fp32,
// igemm_fp8.hip: fp8 operands with bf16 output).
#pragma once
#include "common.h"

#define MAX_TAPS 52
struct Tap { int8_t dy, dx; int16_t widx; };

// One launch covers up to 4 independent sub-problems ("phases") that share A, B, D and the tile shape: the stride^2
// output phases of a strided dgrad / ConvTranspose forward, each a unit-stride gather over its own tap subset.
// Logical tile id = tile_in_phase * nphase + phase, so every XCD gets the same mix of light and heavy phases.
struct Phase { int OHp, OWp, out_oy, out_ox, M, ntaps, tap0, ntm; };
struct GatherArgs {
  const void* A; const void* B; void* D;
  const float* bias; const void* residual; const float* scale;
  int Hi, Wi, Ci;
  int in_sy, in_sx;
  int Ho, Wo;
  int out_sy, out_sx;
  int Nout, ldb, ldd;
  int cshift;
  int accumulate;
  int nphase, ntn, ntiles;     // ntiles = nphase * max_phase(ntm) * ntn
  int hw;                      // heat-map output mode: pixels per image
  int lw;                      // KW3: log2(min(W, 128))
  unsigned a_bytes, b_bytes;
  size_t stat_bytes;           // (host) capacity of stat_partial
  int stat_slices;             // (host) slices the launch writes: nphase * ntm, 0 when the statistics were not fused
  // BatchNorm BACKWARD reduction fused into the epilogue: this launch produces dy of a BatchNorm whose input was bnb_x
  // (same shape as D); per m-tile slice and channel it leaves (sum dy_eff, sum dy_eff * xhat) in bnb_partial[slice][Nout][2].
  // bnb_relu: 0 none, 1 mask from bnb_y > 0, 2 mask recomputed from bnb_x (see bn.hip).
  const void* bnb_x; const void* bnb_y;
  const float* bnb_mean; const float* bnb_invstd; const float* bnb_gamma; const float* bnb_beta;
  float* bnb_partial; int bnb_relu;
  float* stat_partial;         // BatchNorm statistics of the OUTPUT fused into the epilogue: [m-tile slice][Nout][n, mean, M2]
  Phase ph[4];
  Tap taps[MAX_TAPS];
  const float* scale2;         // fp8 path: further device scalars multiplied into the output (operand descales)
  const float* scale3;
  int a_fmt;                   // fp8 path: format of the gathered operand, 0 = e4m3, 1 = e5m2
  // accumulate = 1 only: the value already in D is kept where its bit is set ([rows][ldd / chunk] bytes, bit e = channel
  // chunk * chunk_size + e: the ReLU bit mask of BatchNorm's forward) -- D + this launch's result = masked fork gradient
  const unsigned char* acc_mask;
  int relu;                    // max(0, .) on the finished value (inference: conv + folded BatchNorm + ReLU in one launch)
  // Concatenated-K forward (CAT builds of the gather kernel): D += A2 * B2^T as ONE more K tile behind the conv's own taps --
  // A2 [M][c2] lives at the OUTPUT resolution (row m = output pixel m: single phase, unit output stride), B2 [Nout][c2],
  // c2 <= one K tile.  `heatmap_conv(y) + feature_conv(f)` of the multiscale-fusion heads as one GEMM (regda_7.py:4573-4581).
  const void* A2; const void* B2; const float* bias2;
  int c2; unsigned a2_bytes, b2_bytes;
  int pg_nadd;                 // (pgemm.hip, ADD build) slots of the addend ring
};

// 16-byte chunk with the elements whose mask bit is clear set to zero (bit e = element e; bf16: two elements per word)
template <typename T> __device__ __forceinline__ uint4 keep_masked(uint4 q, unsigned mb) {
  unsigned w[4] = {q.x, q.y, q.z, q.w};
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] &= ((0u - ((mb >> (2 * i)) & 1u)) & 0x0000ffffu) | ((0u - ((mb >> (2 * i + 1)) & 1u)) & 0xffff0000u);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] &= 0u - ((mb >> i) & 1u);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// swizzled byte offset of 16-byte chunk `c` (0..7) in 128-byte row `r`
__device__ __forceinline__ int swz128(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// XCD-aware bijective remap of the linear block id (blocks b and b+8 share an XCD / L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  int start = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + i;
}

// 16-byte load through a buffer descriptor: out-of-range offsets (>= num_records) return zeros, so halo / tail
// handling needs no branch and no zero-initialised destination.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define OOB_OFF ((int)0x80000000)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}


// Persistent pipelined GEMM for 1x1 / unit-stride convs (pgemm.hip): `a` filled exactly as for dispatch_gather.
bool pgemm_eligible(const GatherArgs& a, int elem_size);
int dispatch_pgemm(GatherArgs& a, hipStream_t st);

// fp8-operand build of the gather GEMM (igemm_fp8.hip).  `a` is filled exactly as for the bf16 kernel (element = byte).
int dispatch_gather_fp8(GatherArgs& a, hipStream_t st);

// out[i] (+)= sum over S fp32 slabs of n elements, `stride` elements apart (igemm.hip; also used by wgrad_fp8.hip)
void launch_slab_reduce(const float* ws, float* dw, long n, int S, long stride, int accumulate, hipStream_t st);
