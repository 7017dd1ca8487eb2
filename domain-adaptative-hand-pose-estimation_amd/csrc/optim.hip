// SGD (momentum, weight decay, Nesterov) over a flat fp32 range + the low-precision parameter copy,
// and the plain cast kernel.  Pure streaming: float4 per lane.
#include "common.h"

// torch.optim.SGD semantics (train1.py:141-148): g' = g + wd*p; buf = mu*buf + g' (buf starts at 0, which equals
// torch's "first step: buf = g'"); p -= lr * (nesterov ? g' + mu*buf : buf)
template <typename L>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n,
                                                   const float* __restrict__ lr_dev, float mu, float wd, int nesterov, L* __restrict__ lowp) {
  const float lr = *lr_dev;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 bv = reinterpret_cast<float4*>(buf)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}; const float gg[4] = {gv.x, gv.y, gv.z, gv.w}; float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = gg[e] + wd * pp[e];
      bb[e] = mu * bb[e] + d;
      pp[e] -= lr * (nesterov ? d + mu * bb[e] : bb[e]);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(buf)[i] = make_float4(bb[0], bb[1], bb[2], bb[3]);
    if (lowp) {
#pragma unroll
      for (int e = 0; e < 4; ++e) lowp[4 * i + e] = (L)pp[e];
    }
  }
  // tail
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = g[i] + wd * p[i];
    const float b = mu * buf[i] + d; buf[i] = b;
    const float np = p[i] - lr * (nesterov ? d + mu * b : b); p[i] = np;
    if (lowp) lowp[i] = (L)np;
  }
}

extern "C" int mi355_sgd_nesterov(float* p, const float* g, float* buf, long n, const float* lr_dev, float momentum, float wd,
                                  int nesterov, void* p_lowp, void* stream) {
  if (!p || !g || !buf || !lr_dev || n < 1) MI_FAIL(MI355_EINVAL, "sgd: bad args");
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) MI_FAIL(MI355_EINVAL, "sgd: pointers must be 16-byte aligned");
  int grid = (int)((n / 4 + 255) / 256); if (grid < 1) grid = 1; if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(sgd_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), p, g, buf, n, lr_dev, momentum, wd, nesterov, (bf16_t*)p_lowp);
  MI_CHECK_LAUNCH("sgd");
  return MI355_OK;
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (T)in[i];
}
extern "C" int mi355_cast_f32(const float* in, void* out, long n, int dtype, void* stream) {
  if (!in || !out || n < 1) MI_FAIL(MI355_EINVAL, "cast: bad args");
  int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
  if (dtype == MI355_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), in, (bf16_t*)out, n);
  else if (dtype == MI355_F32) hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), in, (float*)out, n);
  else MI_FAIL(MI355_EINVAL, "cast: bad dtype");
  MI_CHECK_LAUNCH("cast");
  return MI355_OK;
}
