// probe: 4-byte-aligned global_load_dwordx4 (flat) correctness on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, float* o) {
  const int t = threadIdx.x;
  const float4 v = *(const float4*)(p + 1 + 5 * t);   // 4-byte aligned only
  o[t * 4 + 0] = v.x; o[t * 4 + 1] = v.y; o[t * 4 + 2] = v.z; o[t * 4 + 3] = v.w;
}
int main() {
  float *p, *o; const int n = 1024;
  hipMalloc(&p, n * 4); hipMalloc(&o, n * 4);
  float h[n]; for (int i = 0; i < n; ++i) h[i] = i;
  hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o);
  float r[256]; hipError_t e = hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("err=%d:", (int)e); for (int t = 0; t < 12; ++t) printf(" %g", r[t]); printf("  (expect 1 2 3 4 6 7 8 9 11 12 13 14)\n");
  return 0;
}
