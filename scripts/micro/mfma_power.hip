// What the matrix pipe SUSTAINS on this part, by operand data and by MFMA shape.  Round 6 found the convolution kernels
// clock- (power-) limited, not schedule-limited: the same instruction stream runs 26 % faster on all-zero operands than on
// real data, and a kernel with MORE matrix-pipe-busy cycles runs at a LOWER clock.  This probe separates the two factors:
//   * data: LDS filled with (a) zeros, (b) the trivial pattern round 5's ceiling probe used ((i & 7) / 8 as fp32: half of
//     the bf16 values are 0, the rest have one to three mantissa bits), (c) normal random values as bf16 hi | lo pairs
//     (what the kernels multiply);
//   * shape: v_mfma_f32_32x32x16_bf16 (12 per step) against v_mfma_f32_16x16x32_bf16 (24 per step), the same FLOP per
//     step, the same eight ds_read_b128 fragment reads per step, two 256-thread workgroups per CU.
// Prints ms, TFLOP/s (dense bf16) and the in-kernel clock (s_memtime against the 100-MHz s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/mfma_power scripts/micro/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define LDSB 65536

template <bool S16, bool READS>
__global__ __launch_bounds__(256, 2) void kern(const unsigned* __restrict__ src, float* out, unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < LDSB / 4; i += 256) ((unsigned*)smem)[i] = src[i];
  __syncthreads();
  f32x16 acc32[4];
  f32x4 acc16[16];
  for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc32[a][i] = 0.f;
  for (int a = 0; a < 16; ++a) for (int i = 0; i < 4; ++i) acc16[a][i] = 0.f;
  bf16x8 f[8];
  for (int k = 0; k < 8; ++k) f[k] = *(const bf16x8*)(smem + ((w * 64 + lane) * 16 + k * 4096) % LDSB);
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (READS) {
      const int base = ((it * 8) & 31) * 1024 + (w * 64 + lane) * 16;   // conflict-free: a wave reads 1 KiB runs
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = *(const bf16x8*)(smem + ((base + k * 4096 + w * 16384) & (LDSB - 1)));
    }
    if (!S16) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc32[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 + a], f[4 + b], acc32[a * 2 + b], 0, 0, 0);
          acc32[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[a], f[6 + b], acc32[a * 2 + b], 0, 0, 0);
          acc32[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[a], f[4 + b], acc32[a * 2 + b], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc16[a * 4 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[a], f[4 + b], acc16[a * 4 + b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc16[a * 4 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[4 + a], f[b], acc16[a * 4 + b], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) s += acc32[a][0] + acc32[a][15];
  for (int a = 0; a < 16; ++a) s += acc16[a][0] + acc16[a][3];
  asm volatile("s_nop 0" ::"v"(s));
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 123.456f) out[0] = s;
  if (tid == 0) { atomicAdd(clk, c1 - c0); atomicAdd(clk + 1, r1 - r0); }
}

template <bool S16, bool READS>
static void run(const char* what, const unsigned* dsrc, int iters) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, 16);
  hipFuncSetAttribute((const void*)kern<S16, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kern<S16, READS><<<512, 256, LDSB>>>(dsrc, out, clk, iters);   // settle the clock
  hipMemset(clk, 0, 16);
  hipEventRecord(e0);
  kern<S16, READS><<<512, 256, LDSB>>>(dsrc, out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  const double flop = 2048.0 * iters * 12.0 * 32768.0;   // 2048 waves x (12 x 32x32x16 or 24 x 16x16x32) per step
  printf("  %-34s %8.3f ms  %6.0f TFLOP/s bf16 (= %4.0f bf16x3-algorithmic)  in-kernel clock %.2f GHz\n", what, ms, flop / (ms * 1e-3) / 1e12,
         flop / (ms * 1e-3) / 3e12, (double)c[0] / (double)c[1] * 0.1);
  hipFree(out); hipFree(clk);
}

int main() {
  std::vector<unsigned> h(LDSB / 4);
  unsigned* d; hipMalloc(&d, LDSB);
  std::mt19937 rng(7);
  std::normal_distribution<float> N(0.f, 1.f);
  const int iters = 40000;
  for (int mode = 0; mode < 3; ++mode) {
    for (size_t i = 0; i < h.size(); ++i) {
      if (mode == 0) h[i] = 0u;
      else if (mode == 1) { const float v = (float)(i & 7) * 0.125f; h[i] = *(const unsigned*)&v; }
      else {   // two bf16 per word: hi parts and lo parts of normal values, as the kernels' records hold them
        auto bf = [](float v) { unsigned u = *(unsigned*)&v; u += 0x7fffu + ((u >> 16) & 1u); return u >> 16; };
        const float a = N(rng), b = N(rng);
        const bool lo = (i >> 4) & 1;   // alternate 64-byte runs of hi and lo values
        const unsigned ha = bf(a), hb = bf(b);
        const float af = a - *(const float*)&(const unsigned&)(ha << 16), bfv = b - *(const float*)&(const unsigned&)(hb << 16);
        h[i] = lo ? (bf(af) | (bf(bfv) << 16)) : (ha | (hb << 16));
      }
    }
    hipMemcpy(d, h.data(), LDSB, hipMemcpyHostToDevice);
    printf("operand data: %s\n", mode == 0 ? "zeros" : mode == 1 ? "round 5's probe pattern ((i & 7) / 8 as fp32 words)" : "normal random values, bf16 hi | lo");
    run<false, false>("32x32x16, operands in registers", d, iters);
    run<true, false>("16x16x32, operands in registers", d, iters);
    run<false, true>("32x32x16, 8 ds_read_b128 per step", d, iters);
    run<true, true>("16x16x32, 8 ds_read_b128 per step", d, iters);
  }
  return 0;
}
