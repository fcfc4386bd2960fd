#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, float* o, int nbytes) {
  auto r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  const int t = threadIdx.x;
  auto a = __builtin_amdgcn_raw_buffer_load_b128(r, t * 16, 0, 0);
  o[t * 4 + 0] = __builtin_bit_cast(float, a[0]); o[t * 4 + 1] = __builtin_bit_cast(float, a[1]);
  o[t * 4 + 2] = __builtin_bit_cast(float, a[2]); o[t * 4 + 3] = __builtin_bit_cast(float, a[3]);
}
int main() {
  float *p, *o; const int n = 1024;
  hipMalloc(&p, n * 4); hipMalloc(&o, n * 4);
  float h[n]; for (int i = 0; i < n; ++i) h[i] = i + 1;
  hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, n * 4);
  float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int t = 0; t < 12; ++t) printf(" %g", r[t]); printf("\n");
  return 0;
}
