#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) int v2i;
__global__ void probe(unsigned* out, int shift) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
  const int t = threadIdx.x;
  for (int i = t; i < 4096; i += 64) lds[i] = (unsigned char)((i >> shift) & 0xff);
  __syncthreads();
  typedef __attribute__((address_space(3))) v2i* lp;
  v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(lds + t * 8));
  out[t * 2] = (unsigned)v.x; out[t * 2 + 1] = (unsigned)v.y;
}
int main() {
  unsigned *d, h0[128], h1[128];
  hipMalloc(&d, 512);
  probe<<<1, 64>>>(d, 0); hipMemcpy(h0, d, 512, hipMemcpyDeviceToHost);
  probe<<<1, 64>>>(d, 8); hipMemcpy(h1, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int b = 0; b < 8; ++b) {
      unsigned lo = (h0[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff, hi = (h1[l * 2 + b / 4] >> (8 * (b % 4))) & 0xff;
      unsigned a = hi * 256 + lo;
      printf("  [l%2u b%u]", a / 8, a % 8);
    }
    printf("\n");
  }
  return 0;
}
