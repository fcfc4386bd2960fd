// What the tap loop of igemm_pipe_kernel can sustain on its own: per (tap, k-step) a wave reads 8 fragments from LDS
// (ds_read_b128 over 144-byte records: A = 2 row blocks x hi / lo, B = 2 pixel blocks x hi / lo) and issues 12 MFMAs
// (2 x 2 blocks x 3 bf16x3 products).  No global memory, no barriers inside the loop, two 256-thread workgroups per CU
// (77 KB of LDS each, as the production plan).  Prints the MFMA rate as a fraction of 4 x 256 SIMDs x (1 MFMA / 32 clk).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/taploop_ceiling scripts/micro/taploop_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REC 144

template <int READS, int SHADOW = 0>   // SHADOW: bf16 hi / lo splits of fp32 register values interleaved into the MFMA stream (one split = ~5 VALU)
// READS = 8: as shipped; 0: MFMAs only; 4: half the LDS traffic (a 2x larger wave tile would need 12 reads per 24)
__global__ __launch_bounds__(256, 2) void taploop(float* out, unsigned long long* clk, int iters, int taps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 77000 / 4; i += 256) ((float*)smem)[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const unsigned char* X = smem;                 // 340 pixel records
  const unsigned char* W = smem + 340 * REC;     // 3 taps x 64 rows
  float pre[64];
  for (int i = 0; i < 64; ++i) pre[i] = (float)(tid * 64 + i) * 1e-3f;
  unsigned packed[64];
  for (int i = 0; i < 64; ++i) packed[i] = 0;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  const int bbase0 = ((2 * w) * 34 + r) * REC + h * 16, bbase1 = ((2 * w + 1) * 34 + r) * REC + h * 16;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {   // (unrolled: the shadow work indexes registers at compile time)
      const int toff = ((t / 3) * 34 + (t % 3)) * REC;
      const int abase = ((t % 3) * 64 + r) * REC + h * 16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[2], al[2], bh[2], bl[2];
        if (READS >= 8) {
          ah[0] = *(const bf16x8*)(W + abase + ks * 32);            al[0] = *(const bf16x8*)(W + abase + ks * 32 + 64);
          ah[1] = *(const bf16x8*)(W + abase + 32 * REC + ks * 32); al[1] = *(const bf16x8*)(W + abase + 32 * REC + ks * 32 + 64);
          bh[0] = *(const bf16x8*)(X + bbase0 + toff + ks * 32);    bl[0] = *(const bf16x8*)(X + bbase0 + toff + ks * 32 + 64);
          bh[1] = *(const bf16x8*)(X + bbase1 + toff + ks * 32);    bl[1] = *(const bf16x8*)(X + bbase1 + toff + ks * 32 + 64);
        } else if (READS == 4) {
          ah[0] = *(const bf16x8*)(W + abase + ks * 32);            al[0] = *(const bf16x8*)(W + abase + ks * 32 + 64);
          bh[0] = *(const bf16x8*)(X + bbase0 + toff + ks * 32);    bl[0] = *(const bf16x8*)(X + bbase0 + toff + ks * 32 + 64);
          ah[1] = al[0]; al[1] = ah[0]; bh[1] = bl[0]; bl[1] = bh[0];
        } else {
          for (int i = 0; i < 8; ++i) { ah[0][i] = (__bf16)(float)(t + i); al[0][i] = (__bf16)0.5f; bh[0][i] = (__bf16)(float)(ks + i); bl[0][i] = (__bf16)0.25f; }
          ah[1] = al[0]; al[1] = ah[0]; bh[1] = bl[0]; bl[1] = bh[0];
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb], bh[pb], acc[cb][pb], 0, 0, 0);
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bl[pb], acc[cb][pb], 0, 0, 0);
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb], bh[pb], acc[cb][pb], 0, 0, 0);
          }
        if (SHADOW) {
          // SHADOW values per (tap, k-step): affine + hi / lo split + pack, as the input commit does (fma, cvt, sub, cvt, perm)
#pragma unroll
          for (int q = 0; q < SHADOW; ++q) {
            const int idx = ((t * 2 + ks) * SHADOW + q) & 63;
            const float v = fmaf(pre[idx], 1.0001f, 0.5f);
            const unsigned hb = __builtin_bit_cast(unsigned, v) & 0xffff0000u;
            const float lo = v - __builtin_bit_cast(float, hb);
            packed[idx] += (hb >> 16) | (__builtin_bit_cast(unsigned, lo) & 0xffff0000u);
          }
          // 12 MFMAs and 5 * SHADOW VALU of this step, interleaved by the scheduler: 1 MFMA, then a share of the VALU
#pragma unroll
          for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, (5 * SHADOW + 11) / 12, 0);  // VALU
          }
        }
      }
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) s += acc[a][b][0] + acc[a][b][15];
  if (SHADOW) for (int i = 0; i < 64; ++i) s += (float)(packed[i] & 1023u);
  asm volatile("s_nop 0" ::"v"(s));
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (s == 123.456f) out[0] = s;
  if (tid == 0) atomicAdd(clk, c1 - c0);
}

template <int READS, int SHADOW = 0>
static void run(const char* what, int iters, int taps) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, 8); hipMemset(clk, 0, 8);
  hipFuncSetAttribute((const void*)taploop<READS, SHADOW>, hipFuncAttributeMaxDynamicSharedMemorySize, 77632);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  taploop<READS, SHADOW><<<512, 256, 77632>>>(out, clk, 8, taps);
  hipMemset(clk, 0, 8);
  hipEventRecord(e0);
  taploop<READS, SHADOW><<<512, 256, 77632>>>(out, clk, iters, taps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c = 0; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double mfma_per_wave = (double)iters * taps * 2 * 12;
  const double cyc_per_wg = (double)c / 512.0;                       // wave 0 of every workgroup
  // two waves per SIMD share the pipe: busy fraction = 2 x 32 cycles x MFMAs per wave / cycles
  printf("%-28s %8.3f ms  %7.0f cycles/wave  pipe busy %5.1f %%  (%.2f GHz)  %6.0f TFLOP/s bf16 (= %5.0f bf16x3-algorithmic)\n", what, ms, cyc_per_wg,
         100.0 * 2.0 * 32.0 * mfma_per_wave / cyc_per_wg, cyc_per_wg / (ms * 1e6), 2048.0 * mfma_per_wave * 32768.0 / (ms * 1e-3) / 1e12,
         2048.0 * mfma_per_wave * 32768.0 / (ms * 1e-3) / 1e12 / 3.0);
}

int main() {
  run<0>("MFMAs only", 400, 9);
  run<4>("4 reads per 12 MFMAs", 400, 9);
  run<8>("8 reads per 12 MFMAs", 400, 9);
  run<8, 4>("8 reads + 4 splits (20 VALU)", 400, 9);
  run<8, 8>("8 reads + 8 splits (40 VALU)", 400, 9);
  run<8, 12>("8 reads + 12 splits (60 VALU)", 400, 9);
  return 0;
}
