// Probe for ONE HYPOTHESIS about the dropped-addend fault of round 2 (conv dgrad accumulate epilogue, lanes 48-63,
// elements 0 and 2 of a chunk only).  RESULT: NEGATIVE -- see profiles/r03_vcc_sdwa_probe_result.txt.
//
// Hypothesis H1.  The faulty epilogue (rebuilt from the per-float-select source, `hipcc -S`) selected the eight addends with
//     v_cmp_ne_u32_e32 vcc, 0, vBIT ; <two instructions> ; v_cndmask_b32_e32 vW, 0, vW, vcc
// hipcc pads a VALU write of VCC -> VALU read of it as a lane mask to 2 wait states and counts any instruction as one.
// In the faulty code the two states of element 0 were two SDWA instructions and those of element 2 one SDWA + `s_nop 0`
// (an SDWA directly behind the v_cmp in both); element 1 had a plain VOP2 + an SDWA, elements 3..7 `s_nop 1`.  H1: an SDWA
// behind the v_cmp does not count as a wait state on gfx950, so lanes 48-63 of the select see the stale mask.
//
// Test.  Every filler form back to back in inline asm (VCC preset to 0, so a select that sees a stale mask yields 0), plus
// the eight-element block instruction for instruction; 1 / 4 / 16 waves per SIMD; wrong selects counted per lane quarter.
//
// Outcome on MI355X (r03): 0 wrong selects in 6.5e7 .. 1.0e9 selects per lane quarter for EVERY form -- including the
// control with NO instruction between v_cmp and v_cndmask.  The dependency is interlocked in this setting; H1 is not
// supported, and the probe does not reproduce the fault.  What it does not cover: EXEC-masked loops, a preceding
// global_load_sbyte / s_waitcnt, VCC written by SALU and consumed by s_cbranch right before, concurrent LDS / VMEM traffic.
//
//   hipcc --offload-arch=gfx950 -O2 profiles/vcc_sdwa_probe.hip -o /tmp/vcc_probe && /tmp/vcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SDWA(d, s, k) "v_and_b32_sdwa " d ", " s ", " k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
#define PLAIN(d, s, imm) "v_and_b32_e32 " d ", " imm ", " s "\n"

#define PROBE_KERNEL(NAME, FILL)                                                                              \
  __global__ void NAME(const unsigned* __restrict__ m, unsigned* __restrict__ bad, int iters) {              \
    unsigned mb = m[blockIdx.x * blockDim.x + threadIdx.x];                                                   \
    const float v = 1.0f + (float)(threadIdx.x & 63);                                                         \
    const unsigned k2 = 2, k4 = 4;                                                                            \
    unsigned nbad = 0;                                                                                        \
    for (int it = 0; it < iters; ++it) {                                                                      \
      float r; unsigned t0, x, y;                                                                             \
      asm volatile("s_nop 4\n"                                                                                \
                   "s_mov_b64 vcc, 0\n"                                                                       \
                   "v_and_b32_e32 %[t0], 1, %[mb]\n"                                                          \
                   "v_mov_b32_e32 %[x], 0\n"                                                                  \
                   "v_mov_b32_e32 %[y], 0\n"                                                                  \
                   "s_nop 4\n"                                                                                \
                   "v_cmp_ne_u32_e32 vcc, 0, %[t0]\n" FILL                                                    \
                   "v_cndmask_b32_e32 %[r], 0, %[v], vcc\n"                                                   \
                   "s_nop 4\n"                                                                                \
                   : [r] "=&v"(r), [t0] "=&v"(t0), [x] "=&v"(x), [y] "=&v"(y)                                 \
                   : [mb] "v"(mb), [v] "v"(v), [k2] "v"(k2), [k4] "v"(k4)                                     \
                   : "vcc");                                                                                  \
      const float expect = (mb & 1u) ? v : 0.f;                                                               \
      nbad += (r != expect) ? 1u : 0u;                                                                        \
      mb = mb * 1664525u + 1013904223u + x + y;                                                               \
      mb ^= mb >> 13;                                                                                         \
    }                                                                                                         \
    if (nbad) atomicAdd(&bad[threadIdx.x & 63], nbad);                                                        \
  }

PROBE_KERNEL(k_nop1, "s_nop 1\n")
PROBE_KERNEL(k_nop0, "s_nop 0\n")
PROBE_KERNEL(k_none, "")
PROBE_KERNEL(k_sdwa_sdwa, SDWA("%[x]", "%[mb]", "%[k2]") SDWA("%[y]", "%[mb]", "%[k4]"))
PROBE_KERNEL(k_plain_plain, PLAIN("%[x]", "%[mb]", "2") PLAIN("%[y]", "%[mb]", "4"))
PROBE_KERNEL(k_sdwa_nop0, SDWA("%[x]", "%[mb]", "%[k2]") "s_nop 0\n")
PROBE_KERNEL(k_plain_sdwa, PLAIN("%[x]", "%[mb]", "2") SDWA("%[y]", "%[mb]", "%[k4]"))
PROBE_KERNEL(k_sdwa_plain, SDWA("%[x]", "%[mb]", "%[k2]") PLAIN("%[y]", "%[mb]", "4"))
PROBE_KERNEL(k_plain_nop0, PLAIN("%[x]", "%[mb]", "2") "s_nop 0\n")
PROBE_KERNEL(k_sdwa_only, SDWA("%[x]", "%[mb]", "%[k2]"))
PROBE_KERNEL(k_plain_only, PLAIN("%[x]", "%[mb]", "2"))
PROBE_KERNEL(k_sdwa_sdwa_sdwa, SDWA("%[x]", "%[mb]", "%[k2]") SDWA("%[y]", "%[mb]", "%[k4]") SDWA("%[x]", "%[mb]", "%[k4]"))

// the select block of the faulty epilogue, instruction for instruction (register names as operands)
__global__ void k_verbatim(const unsigned* __restrict__ m, unsigned* __restrict__ bad /* [8][64] */, int iters) {
  unsigned mb = m[blockIdx.x * blockDim.x + threadIdx.x];
  const unsigned k1 = 1, k2 = 2, k4 = 4, k8 = 8, k32 = 32, k64 = 64;
  unsigned nbad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    float w0 = 1.f, w1 = 2.f, w2 = 3.f, w3 = 4.f, w4 = 5.f, w5 = 6.f, w6 = 7.f, w7 = 8.f;
    unsigned a16, a19, a24, a25, a35, a36, a37;
    const int sb = (int)(signed char)(mb & 0xff);
    asm volatile("s_nop 4\n"
                 "s_mov_b64 vcc, 0\n"
                 "s_nop 4\n"
                 SDWA("%[a16]", "%[mb]", "%[k1]")
                 SDWA("%[a19]", "%[mb]", "%[k2]")
                 "v_cmp_ne_u32_e32 vcc, 0, %[a16]\n"
                 SDWA("%[a24]", "%[mb]", "%[k4]")
                 SDWA("%[a25]", "%[mb]", "%[k8]")
                 "v_cndmask_b32_e32 %[w0], 0, %[w0], vcc\n"
                 "v_cmp_ne_u32_e32 vcc, 0, %[a19]\n"
                 "v_and_b32_e32 %[a35], 16, %[mb]\n"
                 SDWA("%[a37]", "%[mb]", "%[k64]")
                 "v_cndmask_b32_e32 %[w1], 0, %[w1], vcc\n"
                 "v_cmp_ne_u32_e32 vcc, 0, %[a24]\n"
                 SDWA("%[a36]", "%[mb]", "%[k32]")
                 "s_nop 0\n"
                 "v_cndmask_b32_e32 %[w2], 0, %[w2], vcc\n"
                 "v_cmp_ne_u32_e32 vcc, 0, %[a25]\n"
                 "s_nop 1\n"
                 "v_cndmask_b32_e32 %[w3], 0, %[w3], vcc\n"
                 "v_cmp_ne_u32_e32 vcc, 0, %[a35]\n"
                 "s_nop 1\n"
                 "v_cndmask_b32_e32 %[w4], 0, %[w4], vcc\n"
                 "v_cmp_ne_u32_e32 vcc, 0, %[a36]\n"
                 "s_nop 1\n"
                 "v_cndmask_b32_e32 %[w5], 0, %[w5], vcc\n"
                 "v_cmp_ne_u32_e32 vcc, 0, %[a37]\n"
                 "s_nop 1\n"
                 "v_cndmask_b32_e32 %[w6], 0, %[w6], vcc\n"
                 "v_cmp_gt_i32_e32 vcc, 0, %[sb]\n"
                 "s_nop 1\n"
                 "v_cndmask_b32_e32 %[w7], 0, %[w7], vcc\n"
                 "s_nop 4\n"
                 : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [w4] "+v"(w4), [w5] "+v"(w5), [w6] "+v"(w6),
                   [w7] "+v"(w7), [a16] "=&v"(a16), [a19] "=&v"(a19), [a24] "=&v"(a24), [a25] "=&v"(a25), [a35] "=&v"(a35),
                   [a36] "=&v"(a36), [a37] "=&v"(a37)
                 : [mb] "v"(mb), [sb] "v"(sb), [k1] "v"(k1), [k2] "v"(k2), [k4] "v"(k4), [k8] "v"(k8), [k32] "v"(k32), [k64] "v"(k64)
                 : "vcc");
    const float w[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
#pragma unroll
    for (int e = 0; e < 8; ++e) nbad[e] += (w[e] != (((mb >> e) & 1u) ? (float)(e + 1) : 0.f)) ? 1u : 0u;
    mb = mb * 1664525u + 1013904223u + a35;
    mb ^= mb >> 13;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (nbad[e]) atomicAdd(&bad[e * 64 + (threadIdx.x & 63)], nbad[e]);
}

typedef void (*kern_t)(const unsigned*, unsigned*, int);
static void run(const char* name, kern_t k, const unsigned* dm, unsigned* dbad, int blocks, int iters, int rows) {
  unsigned h[8 * 64];
  hipMemset(dbad, 0, sizeof(h));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dm, dbad, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, dbad, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < rows; ++r) {
    unsigned long q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) q[l / 16] += h[r * 64 + l];
    if (rows > 1) printf("%-18s e=%d", name, r); else printf("%-20s", name);
    printf(" blocks %5d  wrong selects by lane quarter [0-15 16-31 32-47 48-63] = %lu %lu %lu %lu  (of %.3g per quarter)\n", blocks,
           q[0], q[1], q[2], q[3], (double)blocks * 256 / 4 * iters);
  }
}

int main() {
  const int maxblocks = 4096, iters = 4000;
  unsigned* hm = (unsigned*)malloc(maxblocks * 256 * 4);
  srand(7);
  for (int i = 0; i < maxblocks * 256; ++i) hm[i] = (unsigned)rand() * 2654435761u + (unsigned)rand();
  unsigned *dm, *dbad;
  hipMalloc(&dm, maxblocks * 256 * 4); hipMalloc(&dbad, 8 * 64 * 4);
  hipMemcpy(dm, hm, maxblocks * 256 * 4, hipMemcpyHostToDevice);
  const int grids[3] = {256, 1024, 4096};    // 1, 4 and (queued) 16 waves per SIMD
  for (int g = 0; g < 3; ++g) {
    const int b = grids[g];
    run("none (0 states)", k_none, dm, dbad, b, iters, 1);
    run("s_nop 0 (1)", k_nop0, dm, dbad, b, iters, 1);
    run("plain (1)", k_plain_only, dm, dbad, b, iters, 1);
    run("sdwa (1)", k_sdwa_only, dm, dbad, b, iters, 1);
    run("s_nop 1 (2)", k_nop1, dm, dbad, b, iters, 1);
    run("plain,plain (2)", k_plain_plain, dm, dbad, b, iters, 1);
    run("plain,s_nop0 (2)", k_plain_nop0, dm, dbad, b, iters, 1);
    run("plain,sdwa (2)", k_plain_sdwa, dm, dbad, b, iters, 1);
    run("sdwa,plain (2)", k_sdwa_plain, dm, dbad, b, iters, 1);
    run("sdwa,s_nop0 (2)", k_sdwa_nop0, dm, dbad, b, iters, 1);
    run("sdwa,sdwa (2)", k_sdwa_sdwa, dm, dbad, b, iters, 1);
    run("sdwa,sdwa,sdwa (3)", k_sdwa_sdwa_sdwa, dm, dbad, b, iters, 1);
    run("verbatim block", k_verbatim, dm, dbad, b, iters, 8);
  }
  return 0;
}
