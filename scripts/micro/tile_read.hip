// micro-benchmark: HBM read rate of haloed NCHW tile reads (the igemm staging pattern), no compute.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each block: tile TH x TW (+halo 1), C channels; lanes walk pixels; PF = pixel slots per thread
template <int PF, int VEC>
__global__ __launch_bounds__(256) void tile_read(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W,
                                                 int TH, int TW, int tiles_x, int tiles_y, int halo) {
  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  const int txi = bid % tiles_x, tyi = (bid / tiles_x) % tiles_y, n = bid / (tiles_x * tiles_y);
  const int th = TH + 2 * halo, tw = (TW + 2 * halo + VEC - 1) / VEC;   // tw in VEC-wide units
  const int npix = th * tw;
  const int oy0 = tyi * TH - halo, ox0 = txi * TW - halo;
  float acc = 0.f;
  unsigned voff[PF];
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int pix = min(tid + s * 256, npix - 1);
    const int iy = pix / tw, ix = (pix - iy * tw) * VEC;
    const int cy = min(max(oy0 + iy, 0), H - 1), cx = min(max(ox0 + ix, 0), W - VEC);
    voff[s] = (unsigned)(cy * W + (cx & ~(VEC - 1))) * 4u;
  }
  const char* base = (const char*)(x + (long long)n * C * H * W);
  for (int c0 = 0; c0 < C; c0 += 32) {
    float v[PF][32][VEC];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const char* chan = base + (long long)(c0 + j) * H * W * 4;
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        if (VEC == 1) v[s][j][0] = *(const float*)(chan + voff[s]);
        else if (VEC == 2) { float2 t = *(const float2*)(chan + voff[s]); v[s][j][0] = t.x; v[s][j][VEC > 1 ? 1 : 0] = t.y; }
        else { float4 t = *(const float4*)(chan + voff[s]); v[s][j][0] = t.x; v[s][j][VEC > 1 ? 1 : 0] = t.y; v[s][j][VEC > 2 ? 2 : 0] = t.z; v[s][j][VEC > 3 ? 3 : 0] = t.w; }
      }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j)
#pragma unroll
      for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += v[s][j][e];
  }
  if (acc == 123.456f) out[bid] = acc;
}

template <int PF, int VEC>
void run(const char* name, const float* x, float* out, int N, int C, int H, int W, int TH, int TW, int halo) {
  const int tiles_x = W / TW, tiles_y = H / TH;
  const int npix = (TH + 2 * halo) * ((TW + 2 * halo + VEC - 1) / VEC);
  if (npix > PF * 256) { printf("%s: npix %d > %d skip\n", name, npix, PF * 256); return; }
  dim3 grid(N * tiles_x * tiles_y);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((tile_read<PF, VEC>), grid, dim3(256), 0, 0, x, out, C, H, W, TH, TW, tiles_x, tiles_y, halo);
  CK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((tile_read<PF, VEC>), grid, dim3(256), 0, 0, x, out, C, H, W, TH, TW, tiles_x, tiles_y, halo);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double useful = (double)N * C * H * W * 4;
  printf("%-28s TH %3d TW %3d halo %d PF %d VEC %d: %7.3f ms  %6.2f TB/s useful (%d blocks)\n", name, TH, TW, halo, PF, VEC, ms, useful / ms / 1e9, grid.x);
}

int main() {
  const int N = 32, C = 32, H = 256, W = 256;
  float *x, *out; CK(hipMalloc(&x, (size_t)N * C * H * W * 4)); CK(hipMalloc(&out, 1 << 22));
  CK(hipMemset(x, 0, (size_t)N * C * H * W * 4));
  run<2, 1>("8x32 halo (igemm now)", x, out, N, C, H, W, 8, 32, 1);
  run<1, 1>("8x32 no halo", x, out, N, C, H, W, 8, 32, 0);
  run<2, 1>("4x64 halo", x, out, N, C, H, W, 4, 64, 1);
  run<3, 1>("2x128 halo", x, out, N, C, H, W, 2, 128, 1);
  run<3, 1>("1x256 halo", x, out, N, C, H, W, 1, 256, 1);
  run<1, 1>("1x256 no halo", x, out, N, C, H, W, 1, 256, 0);
  run<1, 4>("8x32 no halo vec4", x, out, N, C, H, W, 8, 32, 0);
  run<1, 4>("4x64 no halo vec4", x, out, N, C, H, W, 4, 64, 0);
  run<1, 4>("16x64 no halo vec4", x, out, N, C, H, W, 16, 64, 0);
  run<2, 4>("16x64 halo vec4(aligned)", x, out, N, C, H, W, 16, 64, 1);
  run<1, 2>("8x32 halo vec2", x, out, N, C, H, W, 8, 32, 1);
  run<2, 2>("16x32 halo vec2", x, out, N, C, H, W, 16, 32, 1);
  run<1, 4>("8x32 halo vec4", x, out, N, C, H, W, 8, 32, 1);
  run<1, 4>("16x32 halo vec4", x, out, N, C, H, W, 16, 32, 1);
  return 0;
}
