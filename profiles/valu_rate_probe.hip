// Issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950, measured: NW waves per SIMD each run a long chain-free stream of
// FMAs on 32 independent accumulators; reports FMA lanes per clock per CU.  (Question behind it: is a K loop of scalar FMAs
// worth rewriting with packed ones?  profiles/r03_valu_rate_probe_result.txt)
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x2_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x2_t{(float)threadIdx.x + i, (float)i};
  const f32x2_t x = {a, a + 1.f}, yv = {b, b + 1.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(yv));
        else {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(x[0]), "v"(yv[0]));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][1]) : "v"(x[1]), "v"(yv[1]));
        }
      }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 64 * 1024 * sizeof(float));
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
  printf("%d CUs, %.2f GHz nominal\n", ncu, ghz);
  for (int per = 1; per <= 8; per *= 2)
    for (int pk = 0; pk < 2; ++pk) {
      const int iters = 4000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(ncu * per), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<0>, dim3(ncu * per), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fma = (double)ncu * per * 256 * iters * 4 * 16 * 2;
      printf("%s  %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s  %.1f FMA lanes/clk/CU at nominal clock\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", per, ms,
             2 * fma / ms * 1e-9, fma / (ms * 1e-3) / (ghz * 1e9) / ncu);
    }
  return 0;
}
