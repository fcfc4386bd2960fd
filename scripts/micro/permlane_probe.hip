// probe: what __builtin_amdgcn_permlane32_swap returns on gfx950.  a = 100 + lane, b = 200 + lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(100u + lane, 200u + lane, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
}
int main() {
  unsigned* o; unsigned h[128];
  (void)hipMalloc(&o, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("r0: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
  printf("r1: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
