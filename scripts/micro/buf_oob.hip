// probe: raw buffer load range checking on gfx950 (voffset / soffset / dwordx4 straddling the end)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* p, float* o, int nbytes) {
  auto r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  const int t = threadIdx.x;
  // 0: in range; 1: voffset beyond; 2: huge voffset; 3: soffset pushes beyond, voffset in range; 4: x4 straddling end
  o[0 * 64 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, t * 4, 0, 0));
  o[1 * 64 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, nbytes + t * 4, 0, 0));
  o[2 * 64 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)0xFFFFFFF0u, 0, 0));
  o[3 * 64 + t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, t * 4, nbytes, 0));
  auto q = __builtin_amdgcn_raw_buffer_load_b128(r, nbytes - 8 + (t & 1) * 0, 0, 0);
  o[4 * 64 + t] = __builtin_bit_cast(float, q[t & 3]);
  auto u = __builtin_amdgcn_raw_buffer_load_b128(r, 4 + t * 16, 0, 0);   // 4-byte aligned x4
  o[5 * 64 + t] = __builtin_bit_cast(float, u[3]);
}
int main() {
  float *p, *o; const int n = 1024;
  hipMalloc(&p, n * 4 * 2); hipMalloc(&o, 6 * 64 * 4);
  float h[2 * n]; for (int i = 0; i < 2 * n; ++i) h[i] = i + 1;
  hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, n * 4);
  float r[6 * 64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  const char* names[6] = {"in-range", "voffset>=records", "voffset huge", "soffset beyond (voffset ok)", "x4 straddling end (elems 1022,1023,OOB,OOB)", "x4 at 4B-aligned addr, elem3 (expect 5+4t)"};
  for (int c = 0; c < 6; ++c) { printf("%-50s:", names[c]); for (int t = 0; t < 6; ++t) printf(" %g", r[c * 64 + t]); printf("\n"); }
  return 0;
}
