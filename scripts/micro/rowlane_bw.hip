// What a "lane per row" load pattern streams (the LDS-free weight-gradient kernels are bound by
// it at ~3.8 TB/s).  A tensor [32 images][64 planes][256][256] fp32 (537 MB) is read once by four patterns:
//   k_rowlane   : conv_wgrad3r.hip's: a wave = 32 planes x one 16-pixel piece of a row (lane (r, h): plane r, 32 bytes at
//                 pixel 8 h), two waves of a workgroup on the two halves of a 128-byte line, two on the row halves;
//                 walks down the rows
//   k_rowlane64 : the same with 64 bytes per lane (a wave covers whole 128-byte lines of 32 planes)
//   k_rowlane_x : walks ALONG a row instead of down the rows (consecutive steps read consecutive bytes of a plane)
//   k_stream    : coalesced -- a wave instruction reads 1 KB of one plane row
// usage: hipcc --offload-arch=gfx950 -O2 rowlane_bw.hip -o rowlane_bw && ./rowlane_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 32, C = 64, H = 256, W = 256;

__global__ __launch_bounds__(256) void k_rowlane(const float* __restrict__ x, float* out) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
  const int wx = w & 1, wy = w >> 1;
  const int item = blockIdx.x;                       // (image, 32-px strip, 32-plane block)
  const int cb = item % (C / 32), strip = (item / (C / 32)) % (W / 32), img = item / (C / 32) / (W / 32);
  const float* p = x + ((long long)(img * C + cb * 32 + r) * H + wy * (H / 2)) * W + strip * 32 + 16 * wx + 8 * h;
  f32x4 s = {0, 0, 0, 0};
  for (int y = 0; y < H / 2; ++y) {
    const f32x4 a = *(const f32x4*)(p + (long long)y * W), b = *(const f32x4*)(p + (long long)y * W + 4);
    s += a; s += b;
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_rowlane64(const float* __restrict__ x, float* out) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
  const int item = blockIdx.x;
  const int cb = item % (C / 32), strip = (item / (C / 32)) % (W / 32), img = item / (C / 32) / (W / 32);
  const float* p = x + ((long long)(img * C + cb * 32 + r) * H + w * (H / 4)) * W + strip * 32 + 16 * h;
  f32x4 s = {0, 0, 0, 0};
  for (int y = 0; y < H / 4; ++y) {
    const float* q = p + (long long)y * W;
    s += *(const f32x4*)q; s += *(const f32x4*)(q + 4); s += *(const f32x4*)(q + 8); s += *(const f32x4*)(q + 12);
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_rowlane_x(const float* __restrict__ x, float* out) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
  const int item = blockIdx.x;                       // (image, group of 4 rows... ) each wave one image row, walks along x
  const int cb = item % (C / 32), rg = (item / (C / 32)) % (H / 4), img = item / (C / 32) / (H / 4);
  const float* p = x + ((long long)(img * C + cb * 32 + r) * H + rg * 4 + w) * W + 8 * h;
  f32x4 s = {0, 0, 0, 0};
  for (int k = 0; k < W / 16; ++k) {
    s += *(const f32x4*)(p + k * 16); s += *(const f32x4*)(p + k * 16 + 4);
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

// four waves side by side: 64 pixels = 256 contiguous bytes of a plane row per step, walking down ALL rows
__global__ __launch_bounds__(256) void k_rowlane4(const float* __restrict__ x, float* out) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
  const int item = blockIdx.x;                       // (image, 64-px strip, 32-plane block)
  const int cb = item % (C / 32), strip = (item / (C / 32)) % (W / 64), img = item / (C / 32) / (W / 64);
  const float* p = x + ((long long)(img * C + cb * 32 + r) * H) * W + strip * 64 + 16 * w + 8 * h;
  f32x4 s = {0, 0, 0, 0};
  for (int y = 0; y < H; ++y) {
    const f32x4 a = *(const f32x4*)(p + (long long)y * W), b = *(const f32x4*)(p + (long long)y * W + 4);
    s += a; s += b;
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}
// two image rows per step and wave (the lane's two float4 pairs are rows y and y + 1): 2 x 64 bytes, a 1-KB stride apart
__global__ __launch_bounds__(256) void k_rowlane_2r(const float* __restrict__ x, float* out) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
  const int wx = w & 1, wy = w >> 1;
  const int item = blockIdx.x;
  const int cb = item % (C / 32), strip = (item / (C / 32)) % (W / 32), img = item / (C / 32) / (W / 32);
  const float* p = x + ((long long)(img * C + cb * 32 + r) * H + wy * (H / 2)) * W + strip * 32 + 16 * wx + 8 * h;
  f32x4 s = {0, 0, 0, 0};
  for (int y = 0; y < H / 2; y += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += *(const f32x4*)(p + (long long)(y + k) * W); s += *(const f32x4*)(p + (long long)(y + k) * W + 4); }
  }
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

__global__ __launch_bounds__(256) void k_stream(const float* __restrict__ x, float* out, long long n4) {
  f32x4 s = {0, 0, 0, 0};
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += 256ll * gridDim.x) s += ((const f32x4*)x)[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

int main() {
  const long long n = (long long)N * C * H * W;
  float *x, *out;
  hipMalloc(&x, n * 4); hipMalloc(&out, 64);
  hipMemset(x, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s %8.3f ms  %7.1f GB/s\n", name, ms / 20, n * 4.0 / (ms / 20 * 1e-3) / 1e9);
  };
  run("k_rowlane", [&] { hipLaunchKernelGGL(k_rowlane, dim3(N * (W / 32) * (C / 32)), dim3(256), 0, 0, x, out); });
  run("k_rowlane64", [&] { hipLaunchKernelGGL(k_rowlane64, dim3(N * (W / 32) * (C / 32)), dim3(256), 0, 0, x, out); });
  run("k_rowlane4", [&] { hipLaunchKernelGGL(k_rowlane4, dim3(N * (W / 64) * (C / 32)), dim3(256), 0, 0, x, out); });
  run("k_rowlane_2r", [&] { hipLaunchKernelGGL(k_rowlane_2r, dim3(N * (W / 32) * (C / 32)), dim3(256), 0, 0, x, out); });
  run("k_rowlane_x", [&] { hipLaunchKernelGGL(k_rowlane_x, dim3(N * (H / 4) * (C / 32)), dim3(256), 0, 0, x, out); });
  run("k_stream", [&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, x, out, n / 4); });
  return 0;
}
