// Shared device/host helpers for libpcuda_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pcuda_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))

// ---------------------------------------------------------------- error plumbing
void pcuda_set_error(const char* fmt, ...);
#define PCUDA_FAIL(code, ...)      \
  do {                             \
    pcuda_set_error(__VA_ARGS__);  \
    return (code);                 \
  } while (0)
#define PCUDA_CHECK_LAUNCH(name)                                          \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) PCUDA_FAIL(PCUDA_E_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---------------------------------------------------------------- launch count (prof.hip)
// Every kernel launch of the library goes through hipLaunchKernelGGL: counted here (one relaxed atomic add on the host),
// so that bench.py can report launches per step next to the host's issue time per step (pcuda_launch_count).
extern long long g_pcuda_launches;
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                            \
  do {                                                                 \
    __atomic_fetch_add(&g_pcuda_launches, 1ll, __ATOMIC_RELAXED);      \
    hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);             \
  } while (0)

// ---------------------------------------------------------------- profiling hooks (prof.hip)
// VMEM address rule (round 2, profiles/r02_two_process_determinism.txt).  On a GPU shared by two processes a vector-memory
// load returned wrong data -- for whole 16-lane groups -- when the registers holding ITS ADDRESS were overwritten while it
// was in flight: by its own returning data (the register allocator gives a load's dead address registers to its
// destination, `global_load_dwordx4 v[46:49], v[46:47]`) or by the data of a neighbouring load.  The compiler treats both
// as legal (the target runs without XNACK replay); the evidence says the address is read again.  With the address kept
// alive until the data has been consumed: 0 wrong results in 36000 launches against 20-70 %.  Kernels that can share a
// CU with another process (small LDS footprint: the direct vector-ALU kernels) keep every in-flight load's address
// alive with PCUDA_KEEP after the consuming code; tests/test_isa_rules.py scans their ISA for the pattern.
#define PCUDA_KEEP(ptr) asm volatile("" ::"v"(ptr))

struct ProfScope {
  int fam;
  void* ev0;
  ProfScope(int family, double work, hipStream_t s, const char* tag = nullptr);
  ~ProfScope();
  hipStream_t stream;
};

// hipFuncSetAttribute (the dynamic-LDS opt-in) is per DEVICE while a launcher's guard is per kernel: one bit per device
// ordinal, so a process that drives a second GPU sets it there too (a process-wide flag did not; setting it twice from
// two threads is harmless, hence relaxed atomics).
struct DeviceOnce {
  unsigned long long done = 0;
  // the current device's bit if the attribute still has to be set there, 0 otherwise
  unsigned long long pending() const {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    const unsigned long long b = 1ull << (d & 63);
    return (__atomic_load_n(&done, __ATOMIC_RELAXED) & b) ? 0ull : b;
  }
  void mark(unsigned long long b) { __atomic_fetch_or(&done, b, __ATOMIC_RELAXED); }
};

// ---------------------------------------------------------------- bf16 split helpers
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  bf16x2 p = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32 (round to nearest even, NaN preserving)
  return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ float bf16_hi_as_float(float a) {
  __bf16 h = (__bf16)a;
  return (float)h;
}
// hi/lo packs of two floats: hi = bf16(a), lo = bf16(a - hi).  Written on the packed word (one conversion per
// pair, shift / mask back to fp32): 6 VALU per pair; the per-element form compiled to 8.
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(a, b);
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = pack_bf16x2(a - ha, b - hb);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 32 lanes that share (lane >> 5), on the DPP path (no LDS crossbar, no waits): the total
// lands in lanes 16..31 of each half (lanes 0..15 hold their 16-lane row's sum only)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov0(float v) {   // lanes outside ROW_MASK (and invalid sources) read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float half_wave_sum_hi16(float v) {
  v += dpp_mov0<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov0<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov0<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_mov0<0x140, 0xF>(v);   // row_mirror: every lane of a 16-lane row holds the row's sum
  v += dpp_mov0<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3: lanes 16..31 / 48..63 hold the 32-lane sum
  return v;
}
// sum over each aligned group of N = 16 or 8 lanes (a DPP row or half a row); every lane of the group holds it
template <int N>
__device__ __forceinline__ float row_sum(float v) {
  v += dpp_mov0<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov0<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov0<0x141, 0xF>(v);   // row_half_mirror: lanes 0..7 / 8..15 of a row hold their half's sum
  if (N == 16) v += dpp_mov0<0x140, 0xF>(v);   // row_mirror
  return v;
}
// sum over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
