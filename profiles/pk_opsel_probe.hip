// Stand-alone probe of packed-f32 VALU instructions with register-swapping op_sel on gfx950 (MI355X).
//
// Context ("dropped addend"): ISA-level bisection of the round-2 fault on the real conv kernel showed
// that `v_pk_add_f32 vD, vD, vB op_sel:[0,1] op_sel_hi:[1,0]` (low result = src0.lo + src1.HI) lost its src1 operand in lanes
// 48-63 now and then, while two scalar v_add_f32 on the same registers never did (profiles/dropped_addend_repro.py, variants
// a0 / a1 / a2).  This probe asks whether the instruction misbehaves on its own, and under which neighbours:
//   * which op_sel forms (src1 swapped, src0 swapped, both default, hi-broadcast forms the compiler emits everywhere);
//   * alone on the SIMD, or with a partner wave on the same SIMD issuing MFMAs / LDS reads / global loads meanwhile
//     (in the conv kernel the epilogue of one block runs beside the main loops of the other resident blocks).
// Every result is compared with scalar adds of the same registers; wrong results are counted per lane quarter and half.
//
//   hipcc --offload-arch=gfx950 -O2 profiles/pk_opsel_probe.hip -o /tmp/pk_probe && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

enum { F_SRC1_SWAP = 0, F_SRC0_SWAP, F_DEFAULT, F_HI_BCAST, F_FMA_SRC0_SWAP, F_MUL_SRC0_SWAP, F_SRC1_SWAP_INPLACE, F_SRC0_SWAP_INPLACE, NFORMS };
static const char* form_name[NFORMS] = {"pk_add op_sel:[0,1] op_sel_hi:[1,0]", "pk_add op_sel:[1,0] op_sel_hi:[0,1]",
                                        "pk_add (default)", "pk_add op_sel_hi:[0,1]", "pk_fma op_sel:[1,0,0]",
                                        "pk_mul op_sel:[1,0]", "pk_add d=src0 op_sel:[0,1] op_sel_hi:[1,0]",
                                        "pk_add d=src1 op_sel:[1,0] op_sel_hi:[0,1]"};
enum { P_NONE = 0, P_MFMA, P_LDS, P_VMEM, P_MFMA_AGPR, NPARTNERS };
static const char* partner_name[NPARTNERS] = {"alone", "beside MFMA waves", "beside LDS-read waves", "beside global-load waves",
                                               "beside MFMA waves (AGPR accumulators)"};

template <int FORM>
__device__ __forceinline__ void probe_body(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, int gtid) {
  // per-lane pseudo-random operands, refreshed every iteration; a select in front of the packed op as in the conv epilogue
  unsigned s = (unsigned)gtid * 2654435761u + 12345u;
  unsigned nlo = 0, nhi = 0;
  const float base = in[gtid & 1023];
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const float a0 = base + (float)((s >> 8) & 255), a1 = base - (float)((s >> 16) & 255);
    float b0 = 1.0f + (float)(s & 127), b1 = 3.0f + (float)((s >> 20) & 127);
    const unsigned keep = s >> 30;
    f2 a = {a0, a1}, b, d;
    // the addend pair is produced by v_cndmask (mask bits), like the faulty epilogue
    asm volatile("v_cmp_ne_u32_e32 vcc, 0, %[k0]\n"
                 "s_nop 1\n"
                 "v_cndmask_b32_e32 %[b0], 0, %[b0], vcc\n"
                 "v_cmp_ne_u32_e32 vcc, 0, %[k1]\n"
                 "s_nop 1\n"
                 "v_cndmask_b32_e32 %[b1], 0, %[b1], vcc\n"
                 : [b0] "+v"(b0), [b1] "+v"(b1)
                 : [k0] "v"(keep & 1u), [k1] "v"(keep & 2u)
                 : "vcc");
    b = (f2){b0, b1};
    float e_lo, e_hi;
    if constexpr (FORM == F_SRC1_SWAP) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 1" : "=&v"(d) : "v"(a), "v"(b));
      e_lo = a0 + b1; e_hi = a1 + b0;
    } else if constexpr (FORM == F_SRC0_SWAP) {
      asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]\n s_nop 1" : "=&v"(d) : "v"(a), "v"(b));
      e_lo = b1 + a0; e_hi = b0 + a1;
    } else if constexpr (FORM == F_DEFAULT) {
      asm volatile("v_pk_add_f32 %0, %1, %2\n s_nop 1" : "=&v"(d) : "v"(a), "v"(b));
      e_lo = a0 + b0; e_hi = a1 + b1;
    } else if constexpr (FORM == F_HI_BCAST) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]\n s_nop 1" : "=&v"(d) : "v"(a), "v"(b));
      e_lo = a0 + b0; e_hi = a0 + b1;
    } else if constexpr (FORM == F_FMA_SRC0_SWAP) {
      f2 c = {0.5f, 0.25f};
      asm volatile("v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,0,0]\n s_nop 1" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
      e_lo = __builtin_fmaf(b1, 0.5f, a0); e_hi = __builtin_fmaf(b1, 0.25f, a1);
    } else if constexpr (FORM == F_MUL_SRC0_SWAP) {
      asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0]\n s_nop 1" : "=&v"(d) : "v"(a), "v"(b));
      e_lo = b1 * a0; e_hi = b1 * a1;
    } else if constexpr (FORM == F_SRC1_SWAP_INPLACE) {
      // the form of the faulty epilogue: the destination IS src0 (accumulate in place), src1's halves swapped
      d = a;
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 1" : "+v"(d) : "v"(b));
      e_lo = a0 + b1; e_hi = a1 + b0;
    } else {
      // the variant that cured the kernel (a3): the destination is src1, src0's halves swapped
      d = a;
      asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[0,1]\n s_nop 1" : "+v"(d) : "v"(b));
      e_lo = b1 + a0; e_hi = b0 + a1;
    }
    nlo += (d.x != e_lo) ? 1u : 0u;
    nhi += (d.y != e_hi) ? 1u : 0u;
  }
  const int q = (threadIdx.x & 63) >> 4;
  if (nlo) atomicAdd(&bad[q * 2], nlo);
  if (nhi) atomicAdd(&bad[q * 2 + 1], nhi);
}

// 512-thread blocks: waves 0-3 (one per SIMD) run the packed-op loop, waves 4-7 (their SIMD partners) the chosen partner work
template <int FORM, int PARTNER>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ sink, unsigned* __restrict__ bad, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int t = threadIdx.x, wave = t >> 6;
  for (int i = t; i < 4096; i += 512) lds[i] = in[i & 1023];
  __syncthreads();
  if (wave < 4) {
    probe_body<FORM>(in, bad, iters, blockIdx.x * 256 + t);
  } else if (PARTNER == P_MFMA) {
    f32x16_t acc = {0};
    bf16x8_t x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + t + i); y[i] = (short)(0x3f00 + i); }
    for (int it = 0; it < iters / 2; ++it) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, acc, 0, 0, 0);
    }
    sink[blockIdx.x * 512 + t] = acc[0] + acc[7];
  } else if (PARTNER == P_MFMA_AGPR) {
    // as in the conv kernel: the partner's accumulators live in the accumulator half of the unified register file
    f32x16_t acc0 = {0}, acc1 = {0};
    bf16x8_t x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + t + i); y[i] = (short)(0x3f00 + i); }
    for (int it = 0; it < iters / 2; ++it)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %2, %1" : "+a"(acc0), "+a"(acc1) : "v"(x), "v"(y));
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    sink[blockIdx.x * 512 + t] = acc0[0] + acc1[7];
  } else if (PARTNER == P_LDS) {
    float4 a = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      const float4 q = *reinterpret_cast<const float4*>(&lds[((t * 4 + it * 64) & 4095) & ~3]);
      a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
    }
    sink[blockIdx.x * 512 + t] = a.x + a.y + a.z + a.w;
  } else if (PARTNER == P_VMEM) {
    float a = 0.f;
    for (int it = 0; it < iters / 4; ++it) {
      const float4 q = *reinterpret_cast<const float4*>(&in[((t * 4 + it * 256) & 1023) & ~3]);
      a += q.x + q.y + q.z + q.w;
    }
    sink[blockIdx.x * 512 + t] = a;
  }
}

template <int FORM, int PARTNER>
static void run(const float* din, float* dsink, unsigned* dbad, int blocks, int iters) {
  unsigned h[8];
  hipMemset(dbad, 0, sizeof(h));
  hipLaunchKernelGGL((probe<FORM, PARTNER>), dim3(blocks), dim3(512), 0, 0, din, dsink, dbad, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-38s %-26s blocks %4d  wrong lo/hi by lane quarter: [%u/%u %u/%u %u/%u %u/%u] of %.3g per quarter\n", form_name[FORM],
         partner_name[PARTNER], blocks, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], (double)blocks * 64 * iters);
}

template <int FORM>
static void run_form(const float* din, float* dsink, unsigned* dbad, int iters) {
  const int grids[2] = {256, 1024};
  for (int g = 0; g < 2; ++g) {
    run<FORM, P_NONE>(din, dsink, dbad, grids[g], iters);
    run<FORM, P_MFMA>(din, dsink, dbad, grids[g], iters);
    run<FORM, P_LDS>(din, dsink, dbad, grids[g], iters);
    run<FORM, P_VMEM>(din, dsink, dbad, grids[g], iters);
    run<FORM, P_MFMA_AGPR>(din, dsink, dbad, grids[g], iters);
  }
}

int main() {
  const int iters = 20000;
  float hin[1024];
  srand(3);
  for (int i = 0; i < 1024; ++i) hin[i] = (float)(rand() % 2000) * 0.25f - 250.f;
  float *din, *dsink; unsigned* dbad;
  hipMalloc(&din, sizeof(hin)); hipMalloc(&dsink, 1024 * 512 * 4); hipMalloc(&dbad, 32);
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  run_form<F_SRC1_SWAP>(din, dsink, dbad, iters);
  run_form<F_SRC0_SWAP>(din, dsink, dbad, iters);
  run_form<F_DEFAULT>(din, dsink, dbad, iters);
  run_form<F_HI_BCAST>(din, dsink, dbad, iters);
  run_form<F_FMA_SRC0_SWAP>(din, dsink, dbad, iters);
  run_form<F_MUL_SRC0_SWAP>(din, dsink, dbad, iters);
  run_form<F_SRC1_SWAP_INPLACE>(din, dsink, dbad, iters);
  run_form<F_SRC0_SWAP_INPLACE>(din, dsink, dbad, iters);
  return 0;
}
