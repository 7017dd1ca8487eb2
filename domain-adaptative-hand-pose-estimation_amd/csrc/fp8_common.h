// fp8 helpers shared by the quantisation kernels (igemm_fp8.hip) and the producers that write fp8 copies on the side (bn.hip).
#pragma once
#include "common.h"

// state[0] = scale (x_q = x * scale), state[1] = descale = 1 / scale, state[2] = amax seen since the last update (bits),
// state[3] = the descale of the COPY most recently made with this record (weight packs: the packed copy outlives the scale
// refresh that follows its optimizer step, so its consumers read [3], not [1]).  Activation copies carry their own 16-byte
// record {scale, descale, 0, 0} behind their data instead (mi355_fp8_quantize, write = 2): several copies of one stream are
// alive at once (a forward shared by two backward passes with an optimizer step -- and its scale refresh -- in between).
__device__ __forceinline__ float clamp_fp8(float v, float lim) {   // saturate; NaN stays NaN
  return v != v ? v : fminf(fmaxf(v, -lim), lim);
}
template <bool BF8>
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  constexpr float LIM = BF8 ? 57344.f : 448.f;
  int w = 0;
  if constexpr (BF8) {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(clamp_fp8(a, LIM), clamp_fp8(b, LIM), w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(clamp_fp8(c, LIM), clamp_fp8(d, LIM), w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_fp8(a, LIM), clamp_fp8(b, LIM), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_fp8(c, LIM), clamp_fp8(d, LIM), w, true);
  }
  return (unsigned)w;
}


// amax of a block into state[2]: ONE integer atomic per block (exact and order-independent).  `seen` = state[2] as loaded at the
// START of the kernel (a plain load whose latency hides behind the main loop): blocks that cannot raise it skip the atomic, and
// the atomic itself is issued without a return value, so no block waits for memory at its end -- a dependent load + atomic in
// the tail of every block doubled the time of the short BatchNorm launches, and per-WAVE atomics on one address serialise at
// ~12 ns each (8192 of them made the first quantisation kernel take 100 us whatever the tensor size).
// Every thread of the 256-thread block must call it.
__device__ __forceinline__ unsigned fp8_amax_seen(const float* state) { return reinterpret_cast<const unsigned*>(state)[2]; }
__device__ __forceinline__ void fp8_record_amax(float amax, float* state, unsigned seen) {
  __shared__ float fp8_red_[4];
  amax = block_max<4>(amax, fp8_red_);
  if (threadIdx.x == 0) {
    const unsigned bits = __float_as_uint(amax);
    if (amax > 0.f && bits > seen) (void)atomicMax(reinterpret_cast<unsigned*>(state) + 2, bits);
  }
}
// 8 values * scale -> 8 fp8 bytes
template <bool BF8>
__device__ __forceinline__ uint2 pack8_fp8(const float (&v)[8], float scale) {
  uint2 o;
  o.x = pack4_fp8<BF8>(v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
  o.y = pack4_fp8<BF8>(v[4] * scale, v[5] * scale, v[6] * scale, v[7] * scale);
  return o;
}
