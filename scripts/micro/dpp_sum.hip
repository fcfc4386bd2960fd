#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../pointcloududa_amd/csrc/common.h"
__global__ void k(float* o) {
  const int t = threadIdx.x;
  o[t] = half_wave_sum_hi16((float)t);
  o[64 + t] = half_wave_sum((float)t);
}
int main() {
  float* o; hipMalloc(&o, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  float r[128]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int t = 0; t < 64; ++t) printf("%g ", r[t]); printf("\n");
  printf("ref %g %g\n", r[64], r[64 + 40]);
  return 0;
}
