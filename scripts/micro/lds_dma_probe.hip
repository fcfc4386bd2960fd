// probe: does global_load_lds_dwordx4 reach LDS destinations above 64 KiB on gfx950?  (M0 carries the LDS base)
// writes piece k (1 KiB) of a 150-KiB source to LDS offset k KiB by DMA, copies LDS back out with ds_read; prints the first
// mismatching KiB.   build: hipcc --offload-arch=gfx950 -O2 scripts/micro/lds_dma_probe.hip -o scripts/micro/bin/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define NK 150
__global__ __launch_bounds__(256) void k(const unsigned char* g, unsigned* out) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = w; i < NK; i += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < NK * 256; i += 256) out[i] = ((const unsigned*)smem)[i];
}
int main() {
  const size_t n = NK * 1024;
  std::vector<unsigned> h(n / 4), r(n / 4);
  for (size_t i = 0; i < n / 4; ++i) h[i] = (unsigned)(i * 2654435761u);
  unsigned char* d; unsigned* o;
  hipMalloc(&d, n); hipMalloc(&o, n);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), n, 0, d, o);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(r.data(), o, n, hipMemcpyDeviceToHost);
  int bad = -1;
  for (size_t i = 0; i < n / 4; ++i) if (r[i] != h[i]) { bad = (int)(i / 256); break; }
  printf("status %s; first mismatching KiB: %d (of %d)\n", hipGetErrorString(e), bad, NK);
  return 0;
}
